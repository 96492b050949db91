#!/usr/bin/env python
"""Fuzz the pruned scoring call (deferred-log entry points) over random shapes, both dtypes: knob 3 (candidate keys, the default) must give
the bits of knob 1 (key-per-lane statistics + full column-maximum pass) and of knob 5 (candidate pairs), run to run identical, and stay
within a few steps of knob 0 (the two-pass kernels: other row statistics).   python tools/fuzz_prune.py [n_shapes] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import _lib, ops

lib = _lib.load(); dev = "cuda:0"
n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = torch.cuda.current_stream().cuda_stream


def score(q, k, sink, start, end, knob):
    _, H, q_len, D = q.shape
    _, Hkv, klen, _ = k.shape
    G, m = H // Hkv, end - start
    ws = torch.empty(lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink), dtype=torch.uint8, device=dev)
    ws.random_(0, 255)   # stale bytes in the lists / counters must not matter
    log = torch.empty(Hkv, m, dtype=torch.int32, device=dev)
    out = torch.empty(Hkv, m, dtype=q.dtype, device=dev)
    lib.kvz_debug_set_tunable(b"score_prune", knob)
    try:
        ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), st), "fill")
        ops.check(lib.kvz_score_chunk_log(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, end, q_len, Hkv, G, D,
                                          ops._dtype_code(q.dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel(), st), "score_chunk_log")
        ops.check(lib.kvz_score_finalize_log(log.data_ptr(), log.numel(), out.data_ptr(), ops._dtype_code(q.dtype), st), "finalize")
        torch.cuda.synchronize()
    finally:
        lib.kvz_debug_set_tunable(b"score_prune", -1)
    return out


def bits(t):
    return torch.nan_to_num(t.float(), nan=7.0)


g = torch.Generator(device=dev).manual_seed(321)
bad, worst = 0, 0
for n in range(n_shapes):
    Hkv = rng.choice([1, 2, 4, 8])
    G = rng.choice([1, 2, 4, 7, 8])
    D = rng.choice([64, 128])
    bf = rng.random() < 0.4
    sink = rng.choice([0, 1, 16, 30, 32, 40, 100])
    m = rng.choice([1, 3, 31, 32, 33, 64, 127, 129, 300, 777, 1500, 2000, 2500]) if rng.random() < 0.5 else rng.randint(1, 2600)
    q_len = rng.choice([32, 33, 100, 127, 128, 129, 500, 1013, 2026]) if rng.random() < 0.5 else rng.randint(32, 2100)
    q_len = max(32, min(q_len, 2600 * 8 // (G * 4) + 1))
    start = sink + rng.randint(0, 300)
    end = start + m
    klen = end + rng.randint(0, 300) + q_len
    dtype = torch.bfloat16 if bf else torch.float16
    scale = rng.choice([0.3, 1.0, 2.5])
    q = (torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev) * scale).to(dtype)
    k = (torch.randn(1, Hkv, klen, D, generator=g, device=dev) * scale).to(dtype)
    kind = rng.choice(["plain", "plain", "spike", "copy", "const"])
    if kind == "spike":
        k[0, rng.randrange(Hkv), rng.randrange(klen)] *= 30.0
    if kind == "copy" and q_len >= m:
        kk = k[:, :, start:end].repeat_interleave(G, dim=1)
        q[:, :, :m] = (q[:, :, :m].float() * 0.5 + kk.float() * 0.4).to(dtype)
    if kind == "const":
        q[:] = 0.25; k[:] = 0.5
    a3, a3b = score(q, k, sink, start, end, 3), score(q, k, sink, start, end, 3)
    a1, a5, a0 = score(q, k, sink, start, end, 1), score(q, k, sink, start, end, 5), score(q, k, sink, start, end, 0)
    d0 = (a3.view(torch.int16).int() - a0.view(torch.int16).int()).abs()
    ok = torch.equal(bits(a3), bits(a3b)) and torch.equal(bits(a3), bits(a1)) and torch.equal(bits(a3), bits(a5)) and int(d0.max()) <= 8 \
        and not bool(torch.isnan(a3.float()).any())
    worst = max(worst, int(d0.max()))
    if not ok:
        bad += 1
        print(f"FAIL Hkv={Hkv} G={G} D={D} bf16={bf} sink={sink} start={start} m={m} q_len={q_len} klen={klen} {kind}: run-to-run "
              f"{torch.equal(bits(a3), bits(a3b))} 3==1 {torch.equal(bits(a3), bits(a1))} 3==5 {torch.equal(bits(a3), bits(a5))} vs two-pass worst {int(d0.max())}")
print(f"{n_shapes} shapes: {bad} failures; largest difference to the two-pass call {worst} steps")
