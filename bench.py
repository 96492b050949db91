#!/usr/bin/env python
"""bench.py — KV tokens scored+pruned per second at a 128k-token context, ratio 0.3 (BASELINE.json metric).

One "step" = one full pass of the eviction hot path over one synthetic context that is already resident in HBM:
    --level pair (default, BASELINE configs C2/C3/C4):
        ceil(N/2000) scoring chunks x L layers of  update(repeat K,V) -> _get_score   (attention/score.py:36-65)
        -> global-threshold selection (score.py:88-102) -> compaction of all layers (kvcache.py:152-185)
    --level head (BASELINE config C5, context-independent eviction, model/wrapper.py:40-58):
        head scores [L,Hkv] (the reference's utils/head_score values) -> head-level selection -> compaction of all layers
driven through the drop-in cache object (kvzip_amd.EvictCache), followed (untimed for `value`) by post-prune decode steps.
Weak scaling: every rank (one per GPU) owns one independent context; there is no data-path collective, only the gather of
the per-context result record (kvzip_amd.dist.gather_results) inside the timed region.

    python bench.py [--gpus N --steps K --warmup W]          # N > 1 without a launcher: spawns one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import socket
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GEOM = {  # HF configs of the models BASELINE.json names: layers, query heads, kv heads, head dim
    "qwen2.5-7b": (28, 28, 4, 128),
    "llama3.1-8b": (32, 32, 8, 128),
    "qwen2.5-14b": (48, 40, 8, 128),
    "qwen2.5-0.5b": (24, 14, 2, 64),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
# What the matrix pipe reaches on RANDOM fp16 operands on this part: the package sits at its power limit and the clock follows the
# energy per cycle - v_mfma_f32_32x32x16_f16 back to back on every SIMD runs at 1.65 GHz (19.7 ns per MFMA) instead of 2.21 GHz with
# constant operands (tools/probes/probe_pipe.hip, profiles/r5_scoring_attribution.txt item 5).  Reported beside the spec peak, never instead.
MFMA_RANDOM_DATA_TFLOPS = 1720.0
CLOCK_UNDER_SCORING_GHZ = 1.95  # SQ_WAVE_CYCLES x 4 / waves / launch duration of the scoring kernels (same file, item 1)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="qwen2.5-7b", choices=sorted(GEOM))
    ap.add_argument("--ctx", type=int, default=131072)
    ap.add_argument("--ratio", type=float, default=None, help="keep ratio (default 0.3; 0.6 with --level head)")
    ap.add_argument("--level", default="pair", choices=["pair", "head"],
                    help="pair: score + global threshold (C2-C4); head: context-independent head-level eviction (C5)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--chunk", type=int, default=2000)       # model/wrapper.py:200
    ap.add_argument("--sink", type=int, default=32)
    ap.add_argument("--decode-tokens", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-unfused", action="store_true", help="decode with update / prepare / attend as three calls")
    ap.add_argument("--q-pool", type=int, default=0, help="distinct chunk inputs kept in HBM (0 = all chunks)")
    ap.add_argument("--inputs", default="gauss", choices=["gauss", "copy"],
                    help="gauss (default, the headline): independent Gaussian q / k; copy: repeat-prompt-like - the query (and the re-read key) of position i of "
                         "a chunk resembles the context key of that position, so the softmax rows are peaked as in KVzip's own scoring pass "
                         "(model/wrapper.py:223-249 feeds the context back as a repeat prompt); needs --q-pool 0")
    ap.add_argument("--prof-calls", type=int, default=8,
                    help="the first n scoring calls of every timed step run alone on the caller's stream with their kernels "
                         "bracketed by hipEvents (kernel durations for the roofline; the GPU is idle at a step's start, nothing is "
                         "drained); all other calls overlap on the side streams")
    ap.add_argument("--unfused-update", action="store_true", help="update and _get_score as two library calls")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="set a tuning knob of the library before the run (kvz_debug_set_tunable: measurement only, e.g. flash2_split=0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the RCCL process group even with one GPU and push the result gather and the max-over-ranks "
                         "reduction through it (a 1-GPU box then exercises the collective path of the 8-GPU run)")
    ap.add_argument("--score-streams", type=int, default=0,
                    help="side streams over which the scoring calls of consecutive layers are issued (0 = the library's choice: three with "
                         "the pruned call, two otherwise; 1 = caller's stream)")
    args = ap.parse_args(argv)
    if args.ratio is None:
        args.ratio = 0.6 if args.level == "head" else 0.3
    return args


# ---------------------------------------------------------------------------------------------------------------
# rank logic (one process per GPU).  Backend-agnostic on purpose: tests/test_dist_gloo.py runs exactly this code with
# world_size 2 over gloo on CPU.
# ---------------------------------------------------------------------------------------------------------------
class Ranks:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run or our own spawn)."""

    def __init__(self, expected_world: int, backend: str = "nccl", device=None, force: bool = False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != expected_world:
            raise SystemExit(f"--gpus {expected_world} but WORLD_SIZE={self.world}")
        self.dist = None
        self.forced = bool(force) and self.world == 1
        if self.world > 1 or self.forced:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.forced:
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
            kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self, sync=None):
        if self.dist is not None:
            self.dist.barrier()
        if sync is not None:
            sync()

    def max_over_ranks(self, seconds: float, device="cpu") -> float:
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_records(self, thres: float, r_real: float, len_k: torch.Tensor, layers: int, Hkv: int):
        """The only exchange the path has: one fixed-size record per context (= per rank), library gather."""
        from kvzip_amd.dist import gather_results, pack_record
        return gather_results([pack_record(thres, r_real, len_k)], self.world, layers, Hkv, force_collective=self.forced)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawned(local_rank: int, world: int, port: int, argv, entry):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    entry(argv)


def launch(argv, entry, gpus: int):
    """`python bench.py --gpus N` without torchrun: spawn one process per GPU (rank = local rank) and run `entry` in each."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return entry(argv)
    import torch.multiprocessing as mp
    mp.spawn(_spawned, args=(gpus, _free_port(), argv, entry), nprocs=gpus, join=True)


# ---------------------------------------------------------------------------------------------------------------
def prof_read(lib, name):
    t, c = C.c_double(0), C.c_int64(0)
    lib.kvz_prof_read(name.encode(), C.byref(t), C.byref(c))
    return t.value, c.value


def ulp_keys(t: torch.Tensor) -> torch.Tensor:
    x = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    return torch.where(x >= 0x8000, 0x8000 - (x - 0x8000) - 1, x + 0x8000)


def head_scores_for(model: str, L: int, Hkv: int, dtype) -> torch.Tensor:
    """[L,Hkv] head scores: the reference's own utils/head_score values (fixture tests/golden/g4_head_score.npz) for the
    models it ships them for, seeded random values otherwise."""
    import numpy as np
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "g4_head_score.npz"))
        bits = g[f"{model}/head_score"]
        is_bf16 = bool(g[f"{model}/is_bf16"][0])
        hs = torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16 if is_bf16 else torch.float16)
        if tuple(hs.shape) == (L, Hkv):
            return hs.to(dtype), "reference utils/head_score values (tests/golden/g4_head_score.npz)"
    except (OSError, KeyError):
        pass
    return torch.rand(L, Hkv, generator=torch.Generator().manual_seed(7)).to(dtype), "seeded random head scores"


def cpu_baseline(args, L, H, Hkv, D, dtype, dev, head_scores):
    """Oracle (CPU restatement of the reference path, validated bit-for-bit against the reference's golden vectors) timed on
    this box's host cores on a bounded sample of the same workload.  The same sample doubles as a parity check of the HIP
    scoring kernel at the headline shape (`parity_sample`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kvzip_oracle as orc
    from kvzip_amd import ops
    sink, N, chunk, ratio = args.sink, args.ctx, args.chunk, args.ratio
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    parity = None
    if args.level == "head":
        t0 = time.perf_counter()
        kept, _ = orc.threshold_heads(head_scores, N, ratio)
        valid1 = kept[:1].unsqueeze(1).unsqueeze(-1).expand(1, 1, Hkv, N).contiguous()
        t_sel = time.perf_counter() - t0
        K1 = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dtype)]
        V1 = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dtype)]
        t0 = time.perf_counter()
        orc.prepare_init(K1, V1, valid1, sink)
        t_cmp = time.perf_counter() - t0
        total = t_sel + L * t_cmp
        sample = (f"head-level selection over {L * Hkv} head scores ({t_sel * 1e3:.1f} ms), prepare_init of 1 of {L} layers "
                  f"({t_cmp:.2f} s); extrapolated to the whole context")
    else:
        # the (layer, chunk) calls of the workload come in three shapes (model/wrapper.py:197-221): the first chunk (repeat prompt
        # overhead 13 tokens, window at the sink), the later chunks (overhead 26, window somewhere in the context) and the last,
        # shorter chunk.  One oracle call of each is timed (the extrapolation weighs them by how often they occur) and doubles as the
        # parity check of the HIP kernels on exactly those tensors.
        n_chunks = math.ceil(N / chunk)
        m_last = N - (n_chunks - 1) * chunk
        shapes = [("first", min(chunk, N), 13, 0, 1)]
        if n_chunks > 2:
            shapes.append(("later", chunk, 26, 3 * chunk + 17, n_chunks - 2))
        if n_chunks > 1:
            shapes.append(("last", m_last, 26, 5 * chunk + 3, 1))
        parity = {}
        violations = []
        t_calls, per_shape, d_all, v_all = 0.0, [], [], []
        for tag, m, over, off, count in shapes:
            q_len = m + over
            klen = sink + off + m + q_len
            qf = torch.randn(1, H, q_len, D, generator=g)
            kf = torch.randn(1, Hkv, klen, D, generator=g)
            if args.inputs == "copy":   # (the workload's own kind of inputs: queries and re-read keys resemble the chunk's context keys)
                kc = kf[:, :, sink + off:sink + off + m]
                qf[:, :, :m] = qf[:, :, :m] * 0.5 + kc.repeat_interleave(H // Hkv, dim=1) * 1.5
                kf[:, :, klen - q_len:klen - q_len + m] = kf[:, :, klen - q_len:klen - q_len + m] * 0.3 + kc
            q, k = qf.to(dtype), kf.to(dtype)
            t0 = time.perf_counter()
            want = orc.get_score(q, k, sink, sink + off, sink + off + m)
            dt_call = time.perf_counter() - t0
            t_calls += dt_call * count
            per_shape.append(f"{tag} m={m} q={q_len}: {dt_call:.2f} s x {count}")
            got = ops.score_chunk(q.to(dev), k.to(dev), sink, sink + off, sink + off + m).cpu()
            d_all.append((ulp_keys(got) - ulp_keys(want)).abs().reshape(-1))
            v_all.append((want.reshape(-1), got.reshape(-1)))
            if tag == "first":   # the other dtype once, on the first chunk's tensors
                odt, oname = (torch.bfloat16, "bf16") if dtype == torch.float16 else (torch.float16, "f16")
                ref_o = orc.get_score(qf.to(odt), kf.to(odt), sink, sink + off, sink + off + m)
                got_o = ops.score_chunk(qf.to(odt).to(dev), kf.to(odt).to(dev), sink, sink + off, sink + off + m).cpu()
                do = (ulp_keys(got_o) - ulp_keys(ref_o)).abs()
                vr, tr_o = orc.threshold(ref_o.unsqueeze(0), ratio)
                vg, _ = orc.threshold(got_o.unsqueeze(0), ratio)
                violations += parity_violations(oname, do, ref_o, got_o, vr, vg, tr_o)
                parity[oname] = {"bit_identical": float((do == 0).float().mean()), "within_1ulp": float((do <= 1).float().mean()),
                                 "worst_ulp": int(do.max()), f"mask_hamming@{ratio}": float((vr != vg).float().mean()),
                                 "scores": int(do.numel()), "calls": "first chunk only"}
        d = torch.cat(d_all)
        ref_all, got_all = torch.cat([a for a, _ in v_all]), torch.cat([b for _, b in v_all])
        v_ref, t_ref = orc.threshold(ref_all.view(1, 1, 1, -1), ratio)
        v_got, _ = orc.threshold(got_all.view(1, 1, 1, -1), ratio)
        violations += parity_violations(args.dtype, d, ref_all, got_all, v_ref, v_got, t_ref)
        parity["violations"] = violations   # (a non-empty list makes the bench exit with code 3 after printing its line)
        parity[args.dtype] = {"bit_identical": float((d == 0).float().mean()), "within_1ulp": float((d <= 1).float().mean()),
                              "worst_ulp": int(d.max()), f"mask_hamming@{ratio}": float((v_ref != v_got).float().mean()),
                              "scores": int(d.numel()), "calls": "; ".join(per_shape)}
        parity["shape"] = (f"H{H} Hkv{Hkv} D{D} sink{sink}: one (layer,chunk) call of each shape the workload has (first / later / last "
                           "chunk) vs the CPU oracle on the same tensors")
        n_lc = len(shapes)
        t_lc = t_calls / n_chunks          # average oracle time of a (layer, chunk) call, weighted by how often each shape occurs
        # selection over all L*Hkv*N scores and compaction of ONE layer at full N
        score = (torch.rand(L, 1, Hkv, N, generator=g) ** 8).to(dtype)
        t0 = time.perf_counter()
        valid, _ = orc.threshold(score, ratio)
        t_sel = time.perf_counter() - t0
        K1 = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dtype)]
        V1 = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dtype)]
        t0 = time.perf_counter()
        orc.prepare_init(K1, V1, valid[:1], sink)
        t_cmp = time.perf_counter() - t0
        total = n_chunks * L * t_lc + t_sel + L * t_cmp
        sample = (f"{n_lc} of {n_chunks * L} (layer,chunk) get_score calls at full geometry, one per shape ({'; '.join(per_shape)}; weighted mean {t_lc:.2f} s), threshold "
                  f"over all {L * Hkv * N} scores ({t_sel:.2f} s), prepare_init of 1 of {L} layers ({t_cmp:.2f} s); "
                  "extrapolated to the whole context")
    return {"value": N / total, "unit": "tokens/s", "cores": cores, "kind": "port", "sample": sample}, parity


def main(argv=None):
    args = parse(argv)
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to the C stdout of rank 0 (buffered, it lands after the
    # JSON line when stdout is a file), torch / HIP may warn there as well.  The process's stdout is kept aside for the result line
    # and file descriptor 1 points at stderr from here on.
    result_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    exit_code = 0
    try:
        line = _run(args)
        if line is not None:
            os.write(result_fd, (line + "\n").encode())
            if '"parity_ok": false' in line:   # the bench run is itself a parity gate: the line is out, the exit code says it failed
                print("bench.py: the parity sample violates tests/conftest.py:SCORE_BOUNDS - see parity_sample.violations", file=sys.stderr)
                exit_code = 3
    finally:
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)   # RCCL's banner sits in the C stdio buffer of rank 0: out with it while fd 1 is still stderr
        except OSError:
            pass
        os.dup2(result_fd, 1)   # (an in-process caller - tests, launch() with one GPU - gets its stdout back, also on an exception)
        os.close(result_fd)
    if exit_code:
        sys.exit(exit_code)


def hbm_copy_ceiling(dev, nbytes=1 << 30, reps=5):
    """This box's device-to-device copy rate (read + write bytes per second of a 1-GiB hipMemcpyAsync, best of `reps`): the ceiling
    an HBM-bound kernel that reads and writes in equal parts can reach HERE.  Boxes of this pool differ by up to 15 % on the
    compaction stage with an untouched kernel; the copy rate measured in the same process tells a slow kernel
    from a slow box."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.zero_(); b.copy_(a)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def hbm_copy_kernel_ceiling(lib, dev, nbytes=1 << 30, reps=5):
    """The same 1-GiB copy by a plain 16-bytes-per-lane copy KERNEL (kvz_debug_copy_kernel: ordinary and non-temporal loads / stores, the
    better of the two): the yardstick the micro-architecture guide quotes for achievable HBM bandwidth (6.29 TB/s), where ``copy_`` is the
    runtime's blit path.  A gather that reaches this rate on the same box has no headroom left in the kernel."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.zero_(); b.zero_()
    st = torch.cuda.current_stream().cuda_stream
    best = {}
    for variant, name in ((0, "plain"), (1, "nontemporal")):
        assert lib.kvz_debug_copy_kernel(b.data_ptr(), a.data_ptr(), nbytes, variant, st) == 0
        r = 0.0
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.kvz_debug_copy_kernel(b.data_ptr(), a.data_ptr(), nbytes, variant, st); e1.record(); torch.cuda.synchronize()
            r = max(r, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        best[name] = r
    del a, b
    return best


# bounds of the parity sample: tests/conftest.py:SCORE_BOUNDS (tests/test_host_logic.py keeps the two in step).  dtype: (not identical
# <= a n + b, beyond one step <= c n + d, worst step count)
PARITY_BOUNDS = {"f16": (0.002, 8, 0.0005, 2, 8), "bf16": (0.0005, 4, 0.0001, 2, 8)}


def parity_violations(name, d, want=None, got=None, v_ref=None, v_got=None, thres=None):
    """-> list of messages: the sample's step differences `d` against PARITY_BOUNDS, and (when the masks are given) every flipped mask
    entry must be a non-identical score within the worst deviation of the threshold - the rule of tests/conftest.py:check_mask_flips -
    and there may be at most two of them in a sample of this size."""
    a, b, c, e, w = PARITY_BOUNDS[name]
    n = int(d.numel())
    n_diff, n_far, worst = int((d != 0).sum()), int((d > 1).sum()), int(d.max()) if n else 0
    out = []
    if n_diff > a * n + b or n_far > c * n + e or worst > w:
        out.append(f"{name}: {n_diff} of {n} scores not bit-identical, {n_far} beyond one step, worst {worst} (bounds {a}n+{b}, {c}n+{e}, {w})")
    if v_ref is not None:
        flips = (v_ref.reshape(-1) != v_got.reshape(-1))
        tb = torch.tensor([thres]).to(want.dtype)
        near = (d.reshape(-1) > 0) & ((ulp_keys(want.reshape(-1)) - ulp_keys(tb)).abs() <= max(worst, 1))
        if int((flips & ~near).sum()) or int(flips.sum()) > 2:
            out.append(f"{name}: {int(flips.sum())} mask entries flipped, {int((flips & ~near).sum())} of them away from the threshold")
    return out


def _run(args):
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(args.gpus, "nccl", dev, force=args.force_dist)
    rank, world = ranks.rank, ranks.world

    from kvzip_amd import _lib, ops
    from kvzip_amd.dist import backend_version
    from kvzip_amd.kvcache import EvictCache
    lib = _lib.load()
    for kv_ in args.tune:
        name, val = kv_.split("=")
        assert lib.kvz_debug_set_tunable(name.encode(), int(val)) >= 0, f"unknown knob {name}"

    L, H, Hkv, D = GEOM[args.model]
    dtype = torch.float16 if args.dtype == "f16" else torch.bfloat16
    sink, N, ratio = args.sink, args.ctx, args.ratio
    head_level = args.level == "head"
    # scoring chunks exactly as model/wrapper.py:197-221: 2000-token chunks, repeat prompt overhead 13 / 26 tokens
    chunks = []
    for c, st in enumerate(range(0, N, args.chunk)):
        m = min(args.chunk, N - st)
        chunks.append((sink + st, sink + st + m, m + (13 if c == 0 else 26)))
    q_max = max(c[2] for c in chunks)
    cap = sink + N + q_max + 8

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def randn(*shape):
        return torch.randn(*shape, generator=gen, device=dev, dtype=torch.float32).to(dtype)

    # ---- resident inputs: the prefilled KV of one context and the per-chunk scoring inputs ---------------
    store_k = [torch.empty((1, Hkv, cap, D), dtype=dtype, device=dev) for _ in range(L)]
    store_v = [torch.empty((1, Hkv, cap, D), dtype=dtype, device=dev) for _ in range(L)]
    for l in range(L):
        store_k[l][:, :, :sink + N] = randn(1, Hkv, sink + N, D)
        store_v[l][:, :, :sink + N] = randn(1, Hkv, sink + N, D)
    Qs, Ks, Vs = [], [], []
    pool = 0
    head_scores, head_src = None, None
    if head_level:
        head_scores, head_src = head_scores_for(args.model, L, Hkv, dtype)
        hs_dev = head_scores.to(dev)
    else:
        pool = len(chunks) if args.q_pool <= 0 else min(args.q_pool, len(chunks))
        for p in range(pool):
            Qs.append(randn(L, 1, H, q_max, D))
            Ks.append(randn(L, 1, Hkv, q_max, D))
            Vs.append(randn(L, 1, Hkv, q_max, D))
        if args.inputs == "copy":
            assert pool == len(chunks), "--inputs copy: every chunk needs its own inputs (--q-pool 0)"
            for c, (st, en, q_len) in enumerate(chunks):
                for l in range(L):
                    kc = store_k[l][:, :, st:en]                                  # [1, Hkv, m, D]: the context keys of this chunk
                    Qs[c][l][:, :, :en - st] = (Qs[c][l][:, :, :en - st].float() * 0.5 + kc.repeat_interleave(H // Hkv, dim=1).float() * 1.5).to(dtype)
                    Ks[c][l][:, :, :en - st] = (Ks[c][l][:, :, :en - st].float() * 0.3 + kc.float()).to(dtype)
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    # per (input set, query length) the per-layer views handed to the cache object: what a forward pass would hand over as
    # ready tensors; slicing them inside the timed loop would charge ~15 us of harness indexing to every (layer, chunk)
    views = {}
    for c, (st, en, q_len) in enumerate(chunks):
        key = (c % pool if pool else 0, q_len)
        if pool and key not in views:
            p_ = c % pool
            views[key] = ([Qs[p_][l][:, :, :q_len] for l in range(L)], [Ks[p_][l][:, :, :q_len] for l in range(L)],
                          [Vs[p_][l][:, :, :q_len] for l in range(L)])

    n_prof = max(0, args.prof_calls)
    timing = {"on": False}

    prev_kv = [None]

    def one_step():
        if prev_kv[0] is not None:
            prev_kv[0].close()   # (the previous step's cache: its events go back now, not when the collector gets to it)
        kv = prev_kv[0] = EvictCache(cfg, (sink, sink + N), device=dev, dtype=dtype, verbose=False)
        kv.n_score_streams = max(0, args.score_streams)
        kv.fuse_update_score = not args.unfused_update  # what kvzip_amd.attn / ModelKVzip.scoring do: update + _get_score = one call
        kv.adopt_dense(store_k, store_v, sink + N)
        if head_level:
            # what ModelKVzip.scoring(load_score=True) leaves behind: a stride-0 view of the [L,Hkv] head scores
            kv.score = hs_dev.unsqueeze(-1).expand(-1, -1, N).unsqueeze(1)
        else:
            kv.init_score()
            for c, (st, en, q_len) in enumerate(chunks):
                kv.start_idx, kv.end_idx = st, en          # model/wrapper.py:238-244
                seen = kv._seen_tokens
                Qv, Kv, Vv = views[(c % pool, q_len)]
                # kernel timings: the FIRST `--prof-calls` scoring calls of a step run ALONE on the caller's stream, their kernels
                # bracketed by the library's hipEvents.  The GPU is idle at that point anyway (the previous step ended with the
                # host read of prune), so no pipeline is drained and nothing but the event records is added to the timed region;
                # every other call overlaps on the side streams (brackets there would time kernels that share the GPU)
                n_sample = n_prof if (timing["on"] and c == 0) else 0
                t_c = time.perf_counter()
                for l in range(L):
                    if l < n_sample:
                        kv._score_exclusive = True
                        lib.kvz_prof_enable(1)
                    k_all, _ = kv.update(Kv[l], Vv[l], l)                               # attention/attn.py:44-48
                    kv._get_score(Qv[l], k_all, l)                                      # attention/attn.py:53-54
                    if l < n_sample:
                        lib.kvz_prof_enable(0)
                        kv._score_exclusive = False
                if timing["on"] and c == 1:
                    # pure host cost of an (update, _get_score) pair: the second chunk of a step is enqueued while the launch queues
                    # are still shallow (84 kernels in flight), so the host is not yet throttled by the GPU
                    timing["pair_s"] = timing.get("pair_s", 0.0) + (time.perf_counter() - t_c)
                    timing["pairs"] = timing.get("pairs", 0) + L
                kv.slice(seen)
        kv.start_idx, kv.get_score = sink, False
        timing["issued"] = time.perf_counter()
        if timing["on"]:
            lib.kvz_prof_enable(1)
        thres, r_real = kv.prune(ratio, "head" if head_level else "pair")            # attention/kvcache.py:123-138
        lib.kvz_prof_enable(0)
        return kv, thres, r_real

    for _ in range(args.warmup):
        kv, thres, r_real = one_step()
    lib.kvz_prof_reset()
    timing["on"] = True
    ranks.barrier(torch.cuda.synchronize)
    t0 = time.perf_counter()
    host_issue = 0.0
    for _ in range(args.steps):
        ts = time.perf_counter()
        kv, thres, r_real = one_step()
        host_issue += timing["issued"] - ts  # time the host needed to enqueue the scoring of one context
    # the only exchange of the path (inside the timed region): the result record of every context, library gather
    records = ranks.gather_records(thres, r_real, torch.stack(kv.info["len_k"]), L, Hkv)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0          # this rank's own steps (before it waits for the others)
    ranks.barrier(torch.cuda.synchronize)
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0, dev)
    timing["on"] = False
    assert len(records) == world and all(r is not None and r["n_kept"] > 0 for r in records)

    prof = {n: prof_read(lib, n) for n in ("score_rowstat", "score_bounds", "score_colmax", "select", "select_heads", "compact_gather")}

    # ---- post-prune decode: append + variable-length attention, q_len = 1 (attention only) ------------------
    lib.kvz_prof_reset()
    T = args.decode_tokens
    qd = randn(L, 1, H, 1, D)
    kd, vd = randn(L, 1, Hkv, 1, D), randn(L, 1, Hkv, 1, D)

    def decode_tokens(n):
        for _ in range(n):
            for l in range(L):
                if args.decode_unfused:
                    kf, vf = kv.update(kd[l], vd[l], l)
                    qf, kf, vf, info = kv.prepare(qd[l], kf, vf, l)
                    kv.attend(qf, kf, vf, info)
                else:  # what kvzip_amd/attn.py does for a generation step: append + attention in one launch
                    kv.update_attend(qd[l], kd[l], vd[l], l)
    seen = kv._seen_tokens
    decode_tokens(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    decode_tokens(T)
    torch.cuda.synchronize()
    t_dec = time.perf_counter() - t0
    # kernel durations: a few more tokens with the library's hipEvent brackets (kept out of the timed loop: two event records per
    # launch are host work)
    lib.kvz_prof_reset()
    lib.kvz_prof_enable(1)
    decode_tokens(min(T, 4))
    torch.cuda.synchronize()
    lib.kvz_prof_enable(0)
    # the same step as ONE HIP graph (EvictCache.decode_graph: static input buffers, device-side token counter): what a serving
    # engine replays per token; the per-layer loop above is what a Python forward pass does
    t_graph = None
    if T > 0 and not args.decode_unfused:
        step_graph = kv.decode_graph(qd, kd, vd)
        for _ in range(2):
            step_graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(T):
            step_graph.replay()
        torch.cuda.synchronize()
        t_graph = time.perf_counter() - t0
    attn_ms, attn_n = prof_read(lib, "varlen_attn")
    kept_rows = sum(kv.info["rows_used"])
    len_k_host = kv.info["len_k_host"]
    kv.slice(seen)

    # ---- the same attention call on an AdaKV-style RAGGED layer (head lengths 39 k / 4 k / 120 k / 500 ...): work items are cut
    # from the sum of the lengths, so the GPU fills the same way --------------------------------
    ragged = None
    if rank == 0 and T > 0:
        base = [39000, 4000, 120000, 500]
        lens_r = [base[h % 4] for h in range(Hkv)]
        starts_r, acc = [], 0
        for n_ in lens_r:
            starts_r.append(acc)
            acc += n_ + 1024
        kr_, vr_ = randn(acc, D), randn(acc, D)
        qr_ = randn(Hkv, H // Hkv, D)
        ks_ = torch.tensor(starts_r, dtype=torch.int32, device=dev)
        kl_ = torch.tensor(lens_r, dtype=torch.int32, device=dev)
        ws_ = ops.attn_workspace(Hkv, H // Hkv, 1, D, dev)
        meta_ = ops._meta_host(starts_r, lens_r, Hkv)
        for _ in range(3):
            ops.varlen_attn(qr_, kr_, vr_, ks_, kl_, 1, max(lens_r), workspace=ws_, meta_host=meta_)
        torch.cuda.synchronize()
        lib.kvz_prof_reset()
        lib.kvz_prof_enable(1)
        for _ in range(50):
            ops.varlen_attn(qr_, kr_, vr_, ks_, kl_, 1, max(lens_r), workspace=ws_, meta_host=meta_)
        torch.cuda.synchronize()
        lib.kvz_prof_enable(0)
        r_ms, r_n = prof_read(lib, "varlen_attn")
        r_bytes = 2.0 * sum(lens_r) * D * 2
        if r_n:
            ragged = {"bound": "hbm", "achieved": r_bytes / (r_ms / r_n / 1e3) / 1e9, "unit": "GB/s",
                      "frac": r_bytes / (r_ms / r_n / 1e3) / 1e9 / HBM_PEAK_GBS, "avg_ms": r_ms / r_n, "launches": r_n,
                      "algorithmic_bytes": r_bytes, "head_lengths": lens_r}
        del kr_, vr_

    copy_gbs = hbm_copy_ceiling(dev) if rank == 0 else None
    copyk = hbm_copy_kernel_ceiling(lib, dev) if rank == 0 else None
    copyk_gbs = max(copyk.values()) if copyk else None
    # every rank's own time for the timed steps (N = 1-comparable: a SCALE line can be checked against the BENCH line per rank)
    per_rank_ms = [own_elapsed / args.steps * 1e3]
    if ranks.dist is not None:
        t_all = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        ranks.dist.all_gather(t_all, torch.tensor([per_rank_ms[0]], dtype=torch.float64, device=dev))
        per_rank_ms = [float(t.item()) for t in t_all]
    if rank != 0:
        ranks.close()
        return None

    # ---- roofline of the dominant kernel (live hipEvent timings over the timed region) -----------------------
    def stage(name, work, unit_scale):
        ms, n = prof[name]
        if n == 0:
            return None, None, 0
        avg_s = ms / n / 1e3
        return work / avg_s / unit_scale, ms / n, n

    row_bytes = D * 2
    mask_bytes = L * Hkv * (1 if head_level else N)
    compact_bytes = 2.0 * 2.0 * kept_rows * row_bytes + mask_bytes                           # read+write kept rows of K and V + mask
    c_gbs, c_ms, c_n = stage("compact_gather", compact_bytes, 1e9)
    decode_bytes = 2.0 * (kept_rows + Hkv * L) * row_bytes                                    # every kept K and V row once per token
    attn_gbs = decode_bytes / L / (attn_ms / attn_n / 1e3) / 1e9 if attn_n else None

    # HBM traffic of the kernels, measured offline with PMC counters (separate rocprofv3 --pmc passes, see
    # profiles/*_pmc_traffic.json); only attached when the bench runs the geometry it was measured on
    pmc = {}
    try:
        if args.model == "qwen2.5-7b" and N == 131072 and abs(ratio - 0.3) < 1e-9 and not head_level:
            for name in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
                path = os.path.join(ROOT, "profiles", name)
                if os.path.exists(path):
                    pmc = json.load(open(path))
                    break
    except (OSError, ValueError):
        pmc = {}
    stages = {
        "compact_gather": {"bound": "hbm", "achieved": c_gbs, "unit": "GB/s", "frac": c_gbs / HBM_PEAK_GBS if c_gbs else None,
                           "avg_ms": c_ms, "launches": c_n, "algorithmic_bytes": compact_bytes,
                           "traffic": pmc.get("compact_gather", {}).get("traffic_bytes"),
                           # the same rate against what a 1-GiB device-to-device copy reaches on THIS box in this process
                           "hbm_copy_GBps_this_box": copy_gbs, "frac_of_box_copy": (c_gbs / copy_gbs) if (c_gbs and copy_gbs) else None,
                           # ... and against a plain 16-bytes-per-lane copy KERNEL on this box (the guide's yardstick for achievable HBM bandwidth)
                           "hbm_copy_kernel_GBps_this_box": copyk,
                           "frac_of_box_copy_kernel": (c_gbs / copyk_gbs) if (c_gbs and copyk_gbs) else None},
        "decode_varlen_attn": {"bound": "hbm", "achieved": attn_gbs, "unit": "GB/s",
                               "frac": (attn_gbs / HBM_PEAK_GBS) if attn_gbs else None,
                               "avg_ms": (attn_ms / attn_n) if attn_n else None, "launches": attn_n,
                               "algorithmic_bytes": decode_bytes / L,
                               "traffic": pmc.get("varlen_attn_split", {}).get("traffic_bytes"),
                               # the same stage from the un-bracketed token loop (two hipEvent records per launch pair cost ~3 us)
                               "loop_us_per_layer": (t_dec / T / L * 1e6) if T else None,
                               "frac_loop": (decode_bytes / L / (t_dec / T / L) / 1e9 / HBM_PEAK_GBS) if T else None},
    }
    if ragged is not None:
        stages["decode_varlen_attn_ragged"] = ragged
    if head_level:
        h_ms, h_n = prof["select_heads"]
        stages["select_heads"] = {"bound": "launch", "avg_ms": h_ms / h_n if h_n else None, "launches": h_n,
                                  "algorithmic_bytes": 3.0 * L * Hkv}
        roofline = {"bound": "hbm", "kernel": "compact_gather", "achieved": c_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": c_gbs / HBM_PEAK_GBS, "traffic": None,
                    "note": ("algorithmic bytes = read + write of every kept K and V row (2*2*kept_rows*D*2 B) + one mask byte per "
                             "(layer, head); a dropped head contributes only its sink rows; duration from hipEvents around the "
                             "single gather launch of every timed step")}
        workload = (f"{args.model} geometry (L{L} H{H} Hkv{Hkv} D{D}), {N}-token synthetic context, sink {sink}, --level head "
                    f"(context-independent head scores: {head_src}), ratio {ratio}: head-level select + compact; post-prune "
                    "decode reported beside it; one independent context per GPU")
        metric = "kv_tokens_pruned_per_s"
    else:
        flops_lc = [2.0 * H * D * q * (sink + (en - st) + q) for (st, en, q) in chunks]      # QK^T only
        flops_b = [2.0 * H * D * q * (en - st) for (st, en, q) in chunks]                       # pass-B recompute (ctx columns)
        avg_flops_a, avg_flops_b = flops_lc[0], flops_b[0]   # the bracketed launches are calls of the first chunk (q = m + 13)
        a_tf, a_ms, a_n = stage("score_rowstat", avg_flops_a, 1e12)
        b_tf, b_ms, b_n = stage("score_colmax", avg_flops_b, 1e12)
        if b_ms is None:   # (measurement-only knob values that leave the column-maximum launch out: tools/r6_ab7.sh)
            b_tf, b_ms = 0.0, 0.0
        # the pruned call (default for both dtypes, knob score_prune): two small launches between the passes - merged statistics + group bounds,
        # candidate keys per row group - and a column-maximum pass that recomputes the candidates only (its flops are NOT the full ctx-column flops)
        pruned = prof.get("score_bounds", (0.0, 0))[1] > 0
        m_ms = (prof["score_bounds"][0] / prof["score_bounds"][1]) if pruned else 0.0
        s_gbs, s_ms, s_n = stage("select", 5.0 * L * Hkv * N, 1e9)
        score_combined_tf = avg_flops_a / ((a_ms + m_ms + b_ms) / 1e3) / 1e12
        step_flops = L * sum(flops_lc)                                                       # SURVEY 8(d) flops of one whole step
        stages.update({
            "score_rowstat": {"bound": "mfma", "achieved": a_tf, "unit": "TFLOP/s", "frac": a_tf / MFMA_PEAK_TFLOPS,
                              "avg_ms": a_ms, "launches": a_n,
                              "note": "pass A alone, credited with ALL of the call's 8(d) flops (the convention of rounds 1-4)"},
            "score_colmax": {"bound": "mfma", "achieved": None if pruned else b_tf, "unit": "TFLOP/s", "frac": None if pruned else b_tf / MFMA_PEAK_TFLOPS,
                             "avg_ms": b_ms, "launches": b_n,
                             "note": ("candidate-key pass: per 32-row group only the keys whose column maximum the group can hold are recomputed, 32 gathered keys per MFMA tile (exact bounds from pass A)"
                                      if pruned else "pass B alone over its own recomputed ctx-column flops")},
            "score_bounds": ({"avg_ms": m_ms, "launches": prof["score_bounds"][1],
                              "note": "statistics merge + group bounds, candidate keys per row group (two launches in one bracket)"} if pruned else None),
            "score_combined": {"bound": "mfma", "achieved": score_combined_tf, "unit": "TFLOP/s",
                               "frac": score_combined_tf / MFMA_PEAK_TFLOPS},
            "select": {"bound": "hbm", "achieved": s_gbs, "unit": "GB/s", "frac": s_gbs / HBM_PEAK_GBS, "avg_ms": s_ms,
                       "launches": s_n},
        })
        # the stage north_star prices ("score+prune": selection + compaction): both byte counts over both durations
        if s_ms and c_ms:
            sc_gbs = (5.0 * L * Hkv * N + compact_bytes) / ((s_ms + c_ms) / 1e3) / 1e9
            stages["select_compact"] = {"bound": "hbm", "achieved": sc_gbs, "unit": "GB/s", "frac": sc_gbs / HBM_PEAK_GBS,
                                        "avg_ms": s_ms + c_ms, "algorithmic_bytes": 5.0 * L * Hkv * N + compact_bytes,
                                        "frac_of_box_copy_kernel": (sc_gbs / copyk_gbs) if copyk_gbs else None,
                                        "note": "selection (5 bytes per score: two histogram reads + one mask pass) + plan + gather, over the sum of their "
                                                "bracketed durations"}
        # Round 5: the headline roofline is the scoring STAGE - both launches of a (layer, chunk) call - not pass A alone
        roofline = {
            "bound": "mfma", "kernel": "score (rowstat + merge + bounds + candidate-key colmax)" if pruned else "score (rowstat + colmax)",
            "achieved": score_combined_tf, "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": score_combined_tf / MFMA_PEAK_TFLOPS,
            "avg_ms": a_ms + m_ms + b_ms, "launches": min(a_n, b_n), "pruned_call": pruned,
            "traffic": (sum((pmc.get(kn, {}).get("traffic_bytes") or 0) for kn in
                            (("score_rowstatT2", "score_merge", "score_bounds3", "score_colmax_keys") if pruned else ("score_rowstat", "score_colmax"))) or None),
            "achieved_step_tflops": step_flops / (elapsed / args.steps) / 1e12,   # per GPU: 8(d) flops of one context's step / ms_per_step
            "peak_random_fp16_operands": MFMA_RANDOM_DATA_TFLOPS,
            "frac_of_peak_random_fp16_operands": score_combined_tf / MFMA_RANDOM_DATA_TFLOPS,
            "note": ("the scoring stage: algorithmic flops = 2*H*D*q*(sink+m+q) per (layer,chunk) call (QK^T only, SURVEY §8d) over the "
                     "time of ALL launches of the call (pass A rowstat [+ merge / bounds + candidate keys] + pass B colmax; per launch: roofline_stages); achieved_step_tflops = the "
                     "same flops of a whole step over ms_per_step (side streams, selection and compaction included); kernel durations "
                     f"from hipEvents on the launch stream inside the timed region: the first {n_prof} scoring calls of every step (first "
                     "chunk, q = m + 13) run alone on the caller's stream and are bracketed (the GPU is idle at a step's start: no "
                     f"pipeline is drained), the others overlap on {len({st_.cuda_stream for st_ in getattr(kv, '_score_side', [])}) or 1} side streams; traffic: separate rocprofv3 "
                     "--pmc passes (profiles/*_pmc_traffic.json); peak_random_fp16_operands: what v_mfma_f32_32x32x16_f16 alone "
                     "sustains on random operands at this part's power limit (1.65 GHz; profiles/r5_scoring_attribution.txt)"),
        }
        # What binds the stage is not the matrix pipe: per logit the reference's rounding chain + one exponential are ~5 half-rate VALU
        # instructions beside 1/128 MFMA.  VALU-active cycles per launch (PMC SQ_ACTIVE_INST_VALU quad-cycles x 4, separate rocprofv3
        # --pmc pass, profiles/*_pmc_traffic.json) / 1024 SIMDs / the clock the kernels run at = the time the stage would take if its
        # VALU streams issued back to back with everything else hidden; `frac` = that time / the measured duration.
        sqa, sqb = pmc.get("_sq_counters", {}).get("score_rowstat2", {}), pmc.get("_sq_counters", {}).get("score_colmax3", {})
        if sqa.get("SQ_ACTIVE_INST_VALU") and sqb.get("SQ_ACTIVE_INST_VALU") and not pruned:   # (counters of the two-pass kernels)
            us = lambda sq: 4.0 * sq["SQ_ACTIVE_INST_VALU"] / 1024 / (CLOCK_UNDER_SCORING_GHZ * 1e3)
            mf = lambda sq: 32.0 * sq.get("SQ_INSTS_MFMA", 0) / 1024 / (CLOCK_UNDER_SCORING_GHZ * 1e3)
            mfp = lambda sq: sq.get("SQ_INSTS_MFMA", 0) / 1024 * (32 * 32 * 16 * 2 * 1024 / (MFMA_RANDOM_DATA_TFLOPS * 1e12)) * 1e6
            roofline["valu_issue_bound"] = {
                "valu_active_us": {"rowstat": us(sqa), "colmax": us(sqb)}, "mfma_busy_us": {"rowstat": mf(sqa), "colmax": mf(sqb)},
                "valu_wave_instructions_per_launch": {"rowstat": sqa.get("SQ_INSTS_VALU"), "colmax": sqb.get("SQ_INSTS_VALU")},
                "simds": 1024, "clock_ghz_under_load": CLOCK_UNDER_SCORING_GHZ, "bound_us": us(sqa) + us(sqb),
                "measured_us": (a_ms + b_ms) * 1e3, "frac": (us(sqa) + us(sqb)) / ((a_ms + b_ms) * 1e3),
                # the model that fits every MFMA-heavy kernel of this library within ~5 % (scoring passes, dense forward): at the power
                # limit the matrix pipe and the VALU do not hide each other's ENERGY - time = MFMAs per SIMD x 19.7 ns (what the pipe
                # alone sustains on random fp16 operands, MFMA_RANDOM_DATA_TFLOPS) + VALU-active time
                "mfma_at_power_limit_us": {"rowstat": mfp(sqa), "colmax": mfp(sqb)},
                "valu_plus_mfma_at_power_limit_us": us(sqa) + us(sqb) + mfp(sqa) + mfp(sqb),
                "frac_of_valu_plus_mfma_at_power_limit": (us(sqa) + us(sqb) + mfp(sqa) + mfp(sqb)) / ((a_ms + b_ms) * 1e3),
                "note": "counters from a separate rocprofv3 --pmc pass at this geometry (file read); rounds 1-4 priced this bound at the "
                        "2.35 GHz rocm-smi reports - the kernels run at 1.95 GHz (power limit), where VALU + MFMA time add up to "
                        "~0.95 of the measured duration (profiles/r5_scoring_attribution.txt)"}
        sqs = pmc.get("_sq_counters", {})
        if pruned and all(sqs.get(kn, {}).get("SQ_ACTIVE_INST_VALU") for kn in ("score_rowstatT2", "score_colmax_keys")):
            us = lambda sq: 4.0 * sq["SQ_ACTIVE_INST_VALU"] / 1024 / (CLOCK_UNDER_SCORING_GHZ * 1e3)
            mfp = lambda sq: sq.get("SQ_INSTS_MFMA", 0) / 1024 * (32 * 32 * 16 * 2 * 1024 / (MFMA_RANDOM_DATA_TFLOPS * 1e12)) * 1e6
            ka, kb = sqs["score_rowstatT2"], sqs["score_colmax_keys"]
            roofline["valu_issue_bound"] = {
                "valu_active_us": {"rowstat": us(ka), "colmax_keys": us(kb)}, "mfma_at_power_limit_us": {"rowstat": mfp(ka), "colmax_keys": mfp(kb)},
                "valu_plus_mfma_at_power_limit_us": us(ka) + us(kb) + mfp(ka) + mfp(kb), "measured_us": (a_ms + m_ms + b_ms) * 1e3,
                "frac_of_valu_plus_mfma_at_power_limit": (us(ka) + us(kb) + mfp(ka) + mfp(kb)) / ((a_ms + m_ms + b_ms) * 1e3),
                "simds": 1024, "clock_ghz_under_load": CLOCK_UNDER_SCORING_GHZ,
                "note": "counters of the pruned call's two MFMA kernels from a separate rocprofv3 --pmc pass (file read); time = MFMAs per SIMD x "
                        "19.7 ns + VALU-active time (profiles/r5_scoring_attribution.txt); the rest of the bracket is the latency of the two small "
                        "launches between the passes and of the sparse pass's fixed part, which other streams fill in the loop"}
        workload = (f"{args.model} geometry (L{L} H{H} Hkv{Hkv} D{D}), {N}-token synthetic context, sink {sink}, "
                    f"{len(chunks)} scoring chunks of {args.chunk}, ratio {ratio}: score + select + compact; "
                    "one independent context per GPU")
        metric = "kv_tokens_scored_and_pruned_per_s"

    lens = [x for row in len_k_host for x in row]
    out = {
        "metric": metric, "value": world * N * args.steps / elapsed, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_per_rank": per_rank_ms,   # every rank's own steps (no barrier): comparable with the N = 1 line
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if args.inputs == "gauss" else "synthetic (repeat-prompt-like: --inputs copy)",
        "config": {
            "workload": workload, "level": args.level,
            "ratio": ratio, "real_ratio": r_real, "threshold": thres, "kept_rows": int(kept_rows),
            "head_len_min_max": [int(min(lens)), int(max(lens))],
            "parallelism": f"1 context per GPU x{world}, no data-path collective; result records gathered by "
                           "kvzip_amd.dist.gather_results inside the timed region"
                           + (f" ({backend_version()}; process group of {world} rank(s)" + (", forced on one GPU)" if ranks.forced else ")")
                              if ranks.dist is not None else " (no process group: single rank)"),
            "gathered_contexts": len(records),
            "score_streams": len({st.cuda_stream for st in getattr(kv, "_score_side", [])}) or 1,
            "score_streams_distinct": len({st.cuda_stream for st in getattr(kv, "_score_side", [])}) or 1,
            "host_enqueue_ms_per_step": host_issue / args.steps * 1e3,   # (GPU-bound run: includes the time the host is throttled by full queues)
            "host_us_per_update_score_pair": (timing["pair_s"] / timing["pairs"] * 1e6) if timing.get("pairs") else None,
            "update_score_fused": not args.unfused_update,
            "score_prune": lib.kvz_debug_get_tunable(b"score_prune"),   # 6: the tail of the pruned call pipelined over the calls of a side stream (one launch
                                                                        # per call instead of three); the bracketed calls of roofline_stages run the chained form (3)
            "hbm_copy_GBps_this_box": copy_gbs,   # 1-GiB device-to-device copy (read + write), same process: the box's own HBM ceiling
            "tune": args.tune,
        },
        "roofline": roofline,
        "roofline_stages": stages,
        "decode": {"tokens_per_s": (T / t_dec) if T else None, "ms_per_token": (t_dec / T * 1e3) if T else None, "tokens": T,
                   "ms_per_token_hip_graph": (t_graph / T * 1e3) if t_graph else None,
                   "what": "per token: L x (O(1) append of K,V + variable-length attention), model MLP/projections excluded; "
                           "ms_per_token = the step issued layer by layer from Python (kv.update_attend, what kvzip_amd.attn does), "
                           "ms_per_token_hip_graph = the same step replayed as one HIP graph (EvictCache.decode_graph)"},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], parity = cpu_baseline(args, L, H, Hkv, D, dtype, dev, head_scores)
        if parity is not None:
            out["parity_sample"] = parity
            out["parity_ok"] = not parity.get("violations")
    else:
        out["cpu_baseline"] = None
    ranks.close()
    return json.dumps(out)


if __name__ == "__main__":
    launch(sys.argv[1:], main, parse(sys.argv[1:]).gpus)
