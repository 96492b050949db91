#!/usr/bin/env python
"""Fuzz the pipelined tail of the pruned scoring call (knob score_prune = 6) at the C ABI: random sequences of asynchronous log calls of
DIFFERENT shapes / head dims / dtypes on the same workspace and side stream (so phases of different call shapes sit in one fused launch,
dtype or head-dim changes force a flush, a workspace that is too small for three sets falls back to the chained call), flushed at random
points.  Every call's scores must be the bits of the same call through the chained form (knob 3, synchronous entry point).
   python tools/fuzz_tail.py [n_sequences] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import _lib, ops

lib = _lib.load(); dev = "cuda:0"
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cur = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream(device=dev)
g = torch.Generator(device=dev).manual_seed(11)


def chained(q, k, sink, start, end):
    _, H, q_len, D = q.shape
    _, Hkv, klen, _ = k.shape
    G, m = H // Hkv, end - start
    ws = torch.empty(lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink), dtype=torch.uint8, device=dev)
    log = torch.empty(Hkv, m, dtype=torch.int32, device=dev)
    out = torch.empty(Hkv, m, dtype=q.dtype, device=dev)
    lib.kvz_debug_set_tunable(b"score_prune", 3)
    ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), cur), "fill")
    ops.check(lib.kvz_score_chunk_log(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, end, q_len, Hkv, G, D,
                                      ops._dtype_code(q.dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel(), cur), "score_chunk_log")
    ops.check(lib.kvz_score_finalize_log(log.data_ptr(), log.numel(), out.data_ptr(), ops._dtype_code(q.dtype), cur), "finalize")
    torch.cuda.synchronize()
    return out


def bits(t):
    return torch.nan_to_num(t.float(), nan=7.0)


bad = calls = flushes = fallbacks = 0
for s in range(n_seq):
    n_calls = rng.randint(1, 9)
    handle = lib.kvz_async_create(n_calls)
    assert handle >= 0
    shapes = []
    for c in range(n_calls):
        Hkv = rng.choice([1, 2, 4])
        G = rng.choice([1, 2, 4, 7])
        D = rng.choice([128, 128, 64])
        dtype = torch.bfloat16 if rng.random() < 0.3 else torch.float16
        sink = rng.choice([0, 16, 32, 40])
        m = rng.choice([33, 64, 300, 777, 1500, 2000]) if rng.random() < 0.5 else rng.randint(1, 2100)
        q_len = rng.choice([32, 100, 129, 500, 1013]) if rng.random() < 0.5 else rng.randint(20, 1200)   # (< 32: the two-pass call, never deferred)
        start = sink + rng.randint(0, 200)
        shapes.append((Hkv, G, D, dtype, sink, m, q_len, start))
    need = max(lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink) for (Hkv, G, D, dtype, sink, m, q_len, start) in shapes)
    small = rng.random() < 0.15          # a workspace of ONE set: every call must take the chained form
    ws = torch.empty(need if small else 3 * ((need + 255) // 256 * 256), dtype=torch.uint8, device=dev)
    ws.random_(0, 255)
    fallbacks += small
    lib.kvz_debug_set_tunable(b"score_prune", 6)
    pend = []
    for c, (Hkv, G, D, dtype, sink, m, q_len, start) in enumerate(shapes):
        end = start + m
        klen = end + rng.randint(0, 100) + q_len
        q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).to(dtype)
        k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dtype)
        if rng.random() < 0.1:
            q[0, 0, rng.randrange(q_len), 3] = float("nan")
        log = torch.empty(Hkv, m, dtype=torch.int32, device=dev)
        ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), cur), "fill")
        ops.check(lib.kvz_score_chunk_async_log(handle, c, cur, side.cuda_stream, q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start,
                                                end, q_len, Hkv, G, D, ops._dtype_code(dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel()), "async_log")
        pend.append((q, k, sink, start, end, log))
        calls += 1
        if rng.random() < 0.2 or c == n_calls - 1:
            rc = lib.kvz_score_tail_flush_async(handle, c, ws.data_ptr(), side.cuda_stream)
            assert rc >= 0, rc
            flushes += rc
            ops.check(lib.kvz_async_wait(handle, -1, cur), "wait")
            lib.kvz_debug_set_tunable(b"score_prune", 3)
            for (q2, k2, sink2, start2, end2, log2) in pend:
                out = torch.empty(log2.shape, dtype=q2.dtype, device=dev)
                ops.check(lib.kvz_score_finalize_log(log2.data_ptr(), log2.numel(), out.data_ptr(), ops._dtype_code(q2.dtype), cur), "finalize")
                torch.cuda.synchronize()
                want = chained(q2, k2, sink2, start2, end2)
                if not torch.equal(bits(out), bits(want)):
                    bad += 1
                    print(f"MISMATCH sequence {s} call shape H{q2.shape[1]} q{q2.shape[2]} D{q2.shape[3]} m{end2 - start2} {q2.dtype}: "
                          f"{int((bits(out) != bits(want)).sum())} of {out.numel()} scores differ")
            pend = []
            lib.kvz_debug_set_tunable(b"score_prune", 6)
    assert lib.kvz_score_tail_flush(ws.data_ptr()) == 0   # (nothing may be left pending)
    lib.kvz_async_destroy(handle)
    del ws
lib.kvz_debug_set_tunable(b"score_prune", -1)
print(f"{n_seq} sequences, {calls} calls ({fallbacks} sequences on a one-set workspace), {flushes} flushes that launched something: {bad} mismatching calls")
sys.exit(1 if bad else 0)
