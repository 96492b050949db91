#!/usr/bin/env python
"""Micro-driver for profiling: a few kvz_score_chunk / compaction / decode-attention launches at the bench geometry
(Qwen2.5-7B, one layer).  Used under rocprofv3 --pmc (counter passes are slow, so this keeps the launch count small)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "score"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
H, Hkv, D, sink, N, m = 28, 4, 128, 32, 131072, 2000
q_len = m + 26
klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
dt = torch.bfloat16 if os.environ.get("PROF_DTYPE") == "bf16" else torch.float16
if what == "score":
    q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
    k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
    start = sink + 60000
    # the deferred-log entry point (what a scoring pass calls): the pruned call for fp16 by default, the two-pass call with
    # KVZIP_SCORE_PRUNE=0 in the environment
    from kvzip_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.kvz_score_workspace_bytes(Hkv, H // Hkv, q_len, m, sink), dtype=torch.uint8, device=dev)
    log = torch.empty(Hkv, m, dtype=torch.int32, device=dev)
    def call():
        ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), st), "fill")
        ops.check(lib.kvz_score_chunk_log(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, start + m, q_len, Hkv, H // Hkv, D,
                                          ops._dtype_code(q.dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel(), st), "score_chunk_log")
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    print(f"score_chunk_log: {(time.perf_counter() - t0) / iters * 1e6:.1f} us per call")
elif what == "attn":
    G = H // Hkv
    lens = torch.tensor([39000, 39500, 38800, 39900], dtype=torch.int32, device=dev)
    slack = 1024
    starts = torch.tensor([0, 40024, 40024 * 2, 40024 * 3 + 500], dtype=torch.int32, device=dev)
    rows = 40024 * 4 + 2000
    k = torch.randn(rows, D, generator=g, device=dev).to(dt)
    v = torch.randn(rows, D, generator=g, device=dev).to(dt)
    q = torch.randn(Hkv, G, D, generator=g, device=dev).to(dt)
    ws = ops.attn_workspace(Hkv, G, 1, D, dev)
    for _ in range(2):
        ops.varlen_attn(q, k, v, starts, lens, 1, 39900, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.varlen_attn(q, k, v, starts, lens, 1, 39900, workspace=ws)
    torch.cuda.synchronize()
    print(f"varlen_attn: {(time.perf_counter() - t0) / iters * 1e6:.1f} us per call")
elif what == "flash":
    # the scoring forward's dense attention: 2 026 query positions x 28 heads against 133 k keys (kvz_flash2.hip)
    q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
    k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
    v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
    ops.flash_fwd(q, k, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.flash_fwd(q, k, v)
    torch.cuda.synchronize()
    print(f"flash_fwd: {(time.perf_counter() - t0) / iters * 1e6:.1f} us per call")
elif what == "compact":
    L = 28
    store_k = [torch.randn(1, Hkv, sink + N, D, generator=g, device=dev).to(dt) for _ in range(L)]
    store_v = [torch.randn(1, Hkv, sink + N, D, generator=g, device=dev).to(dt) for _ in range(L)]
    valid = torch.rand(L, 1, Hkv, N, generator=g, device=dev) < 0.3
    for _ in range(iters):
        plan = ops.compact_plan(valid, sink, sink + N, slack=1024)
        totals = (plan.len_k.cpu().sum(-1) + 1024 * Hkv).tolist()
        ko, vo = ops.compact_layers(store_k, store_v, plan, totals)
    torch.cuda.synchronize()
    kept = int(plan.len_k.sum())
    print(f"compact: kept rows {kept}, algorithmic bytes {2 * 2 * kept * D * 2 + L * Hkv * N}")
elif what == "select":
    import ctypes as C
    from kvzip_amd import _lib
    lib = _lib.load()
    score = (torch.rand(28, 1, Hkv, N, generator=g, device=dev) ** 8).to(dt)
    for rnd in range(2):
        for cap, emit in ((256, 112), (256, 256), (256, 512), (256, 1024), (128, 256), (384, 256)):
            lib.kvz_debug_set_tunable(b"sel_blocks", cap); lib.kvz_debug_set_tunable(b"emit_blocks", emit)
            for _ in range(3):
                ops.select_threshold(score, 0.3, row_len=N)
            torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
            for _ in range(iters):
                ops.select_threshold(score, 0.3, row_len=N)
            torch.cuda.synchronize(); lib.kvz_prof_enable(0)
            t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(b"select", C.byref(t), C.byref(c))
            print(f"round {rnd} sel_blocks {cap} (1024 threads) emit_blocks {emit}: select (3 launches, plain path) {t.value / max(c.value, 1) * 1e3:.1f} us", flush=True)
    lib.kvz_debug_set_tunable(b"sel_blocks", -1); lib.kvz_debug_set_tunable(b"emit_blocks", -1)
    print("select: algorithmic bytes", 5 * score.numel())
