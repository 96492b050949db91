#!/usr/bin/env python
"""Fuzz kvz_score_chunk over random shapes: a second, structurally different build of the kernels (e.g. the non-persistent
ones from the start of the round) is the cross-check; every shape is also run twice for bit-identity.
   python tools/fuzz_score.py <lib_under_test.so> <lib_cross_check.so> [n_shapes] [seed]"""
import ctypes as C
import random
import sys

import torch

libs = []
for path in sys.argv[1:3]:
    lib = C.CDLL(path)
    lib.kvz_score_workspace_bytes.restype = C.c_size_t
    lib.kvz_score_workspace_bytes.argtypes = [C.c_int] * 5
    lib.kvz_score_chunk.restype = C.c_int
    lib.kvz_score_chunk.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64] + [C.c_int] * 9 + [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    libs.append(lib)
n_shapes = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = random.Random(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
dev = "cuda:0"
ws = [torch.empty(256 << 20, dtype=torch.uint8, device=dev) for _ in libs]


def run(i, q, k, sink, start, end, dt):
    lib = libs[i]
    _, H, q_len, D = q.shape
    _, Hkv, klen, _ = k.shape
    G, m = H // Hkv, end - start
    out = torch.empty((1, Hkv, m), dtype=q.dtype, device=dev)
    assert ws[i].numel() >= lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink)
    rc = lib.kvz_score_chunk(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, end, q_len, Hkv, G, D, dt,
                             out.data_ptr(), out.stride(1), ws[i].data_ptr(), ws[i].numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


def ulp(a, b):
    ia, ib = a.view(torch.int16).int(), b.view(torch.int16).int()
    return (ia - ib).abs()


g = torch.Generator(device=dev).manual_seed(123)
worst_frac, worst_ulp, bad = 1.0, 0, 0
for n in range(n_shapes):
    Hkv = rng.choice([1, 2, 4, 8])
    G = rng.choice([1, 2, 4, 7, 8])
    D = rng.choice([64, 128])
    bf = rng.random() < 0.3
    sink = rng.choice([0, 1, 16, 30, 32, 40])
    m = rng.choice([1, 3, 31, 64, 127, 128, 129, 300, 777, 1500, 2000, 2500]) if rng.random() < 0.5 else rng.randint(1, 2600)
    q_len = rng.choice([1, 2, 17, 32, 33, 100, 127, 128, 129, 500, 1013, 2026]) if rng.random() < 0.5 else rng.randint(1, 2100)
    q_len = min(q_len, 2600 * 8 // (G * 4) + 1)
    gap_before = rng.randint(0, 300)
    gap_after = rng.randint(0, 300)
    start = sink + gap_before
    end = start + m
    klen = end + gap_after + q_len
    dtype = torch.bfloat16 if bf else torch.float16
    scale = rng.choice([0.3, 1.0, 2.5])
    q = (torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev) * scale).to(dtype)
    k = (torch.randn(1, Hkv, klen, D, generator=g, device=dev) * scale).to(dtype)
    a = run(0, q, k, sink, start, end, 1 if bf else 0)
    a2 = run(0, q, k, sink, start, end, 1 if bf else 0)
    b = run(1, q, k, sink, start, end, 1 if bf else 0)
    torch.cuda.synchronize()
    det = torch.equal(a.view(torch.int16), a2.view(torch.int16))
    d = ulp(a, b)
    frac = float((d == 0).float().mean())
    mx = int(d.max())
    nan = bool(torch.isnan(a.float()).any())
    ok = det and not nan and mx <= 8 and (frac >= 0.97 or a.numel() < 200)
    worst_frac, worst_ulp = min(worst_frac, frac), max(worst_ulp, mx)
    if not ok:
        bad += 1
        print(f"FAIL shape Hkv={Hkv} G={G} D={D} bf16={bf} sink={sink} start={start} m={m} q_len={q_len} klen={klen}: deterministic={det} "
              f"nan={nan} identical={frac:.4f} worst={mx}")
print(f"{n_shapes} shapes: {bad} failures; lowest identical fraction {worst_frac:.4f}, largest difference {worst_ulp} half-ulps")
