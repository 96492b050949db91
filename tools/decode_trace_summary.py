#!/usr/bin/env python
"""Summary of a rocprofv3 --kernel-trace csv of tools/decode_graph_probe.py: durations of the split / combine kernels and the gaps between
consecutive kernels (end of one to start of the next), over the last tokens of the run.   python tools/decode_trace_summary.py <csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
att = [e for e in ev if "varlen_attn" in e[2] or "add_i32" in e[2]]
att = att[-(57 * 16):]   # the last 16 tokens (28 x (split + combine) [+ the counter node of the graph])
def short(n): return "split" if "split" in n else ("combine" if "combine" in n else "add_i32")
dur, gap = {}, {}
for i, (s, e, n) in enumerate(att):
    dur.setdefault(short(n), []).append((e - s) / 1e3)
    if i:
        key = short(att[i - 1][2]) + "->" + short(n)
        gap.setdefault(key, []).append((s - att[i - 1][1]) / 1e3)
mean = lambda v: sum(v) / max(len(v), 1)
for k, v in dur.items(): print(f"  kernel {k:8s} n {len(v):4d} mean {mean(v):6.2f} us")
for k, v in gap.items(): print(f"  gap {k:18s} n {len(v):4d} mean {mean(v):6.2f} us (min {min(v):.2f}, max {max(v):.2f})")
span = (att[-1][1] - att[0][0]) / 1e3
print(f"  span of the sample {span:.1f} us = {span / 16:.1f} us per token; kernels busy {sum(sum(v) for v in dur.values()):.1f} us")
