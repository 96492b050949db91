#!/usr/bin/env python
"""Timeline of the scoring loop from a rocprofv3 --kernel-trace csv (…_kernel_trace.csv): how long the dominant pass-A kernels run, what
sits between the end of one and the start of the next, and which small launches (merge / bounds / sparse pass) run there.
   python tools/timeline_score.py <kernel_trace.csv> [first_row_of_sample]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    for k in ("score_rowstatT2", "score_rowstat2", "score_tail", "score_merge", "score_bounds2", "score_bounds3", "score_colmax_sparse", "score_colmax_keys", "score_colmax3", "dense_append", "finalize"):
        if k in n: return k
    return n.split("(")[0][-28:]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows), key=lambda x: x[0])
A = [e for e in ev if e[2].startswith("score_rowstat")]
# steady state: the middle half of the pass-A launches
lo, hi = len(A) // 4, 3 * len(A) // 4
dur = [(e[1] - e[0]) / 1e3 for e in A[lo:hi]]
period = [(A[i + 1][0] - A[i][0]) / 1e3 for i in range(lo, hi)]
gap = [(A[i + 1][0] - A[i][1]) / 1e3 for i in range(lo, hi)]
ovl = [max(0.0, (min(A[i][1], A[i + 1][1]) - A[i + 1][0])) / 1e3 for i in range(lo, hi)]
mean = lambda v: sum(v) / max(len(v), 1)
print(f"pass A launches {len(A)}; steady-state sample {hi - lo}: duration {mean(dur):.1f} us (min {min(dur):.1f} max {max(dur):.1f}), start-to-start period {mean(period):.1f} us, "
      f"end-to-next-start {mean(gap):.1f} us (negative = overlap), overlap of consecutive launches {mean(ovl):.1f} us")
by = collections.defaultdict(list)
for e in ev:
    if A[lo][0] <= e[0] <= A[hi][0]: by[e[2]].append((e[1] - e[0]) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:24s} n {len(v):5d} mean {mean(v):7.2f} us  total {sum(v) / 1e3:8.2f} ms")
# busy time of the union of pass-A intervals vs wall
t0, t1 = A[lo][0], A[hi][1]
cov, cur_s, cur_e = 0, None, None
for s, e, _, _ in A[lo:hi + 1]:
    if cur_e is None or s > cur_e:
        if cur_e is not None: cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
cov += cur_e - cur_s
print(f"wall {(t1 - t0) / 1e3:.0f} us, some pass-A kernel running {cov / 1e3:.0f} us ({100.0 * cov / (t1 - t0):.1f} %)")
first = int(sys.argv[2]) if len(sys.argv) > 2 else next(i for i, e in enumerate(ev) if e[0] >= A[lo][0])
base = ev[first][0]
print("sample (start us, end us, duration, queue, kernel):")
for s, e, n, q in ev[first:first + 48]:
    print(f"  {(s - base) / 1e3:9.1f} {(e - base) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q:>3s}  {n}")
