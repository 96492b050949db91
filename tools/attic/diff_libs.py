#!/usr/bin/env python
"""Bitwise comparison of kvz_score_chunk between two builds of the library (A/B debugging):
   python tools/diff_libs.py <libA.so> <libB.so> [iters]"""
import ctypes as C
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
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda:0"


def run(lib, q, k, sink, start, end, ws):
    _, H, q_len, D = q.shape
    _, Hkv, klen, _ = k.shape
    G, m = H // Hkv, end - start
    out = torch.empty((1, Hkv, m), dtype=q.dtype, device=dev)
    need = lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink)
    assert ws.numel() >= need
    rc = lib.kvz_score_chunk(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, end, q_len, Hkv, G, D, 0,
                             out.data_ptr(), out.stride(1), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


shapes = [  # (H, Hkv, D, sink, N, start, end, q_len)
    (14, 2, 64, 30, 2048, 30, 2030, 2013), (14, 2, 64, 30, 2048, 2030, 2078, 74),
    (28, 4, 128, 32, 8192, 32 + 4000, 32 + 6000, 2026), (32, 8, 128, 32, 3000, 732, 2732, 2026),
]
g = torch.Generator(device=dev).manual_seed(1)
ws = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in libs]
for H, Hkv, D, sink, N, start, end, q_len in shapes:
    bad = 0
    for it in range(iters):
        q = torch.randn(1, H, q_len, D, generator=g, device=dev).half()
        k = torch.randn(1, Hkv, sink + N + q_len, D, generator=g, device=dev).half()
        a = run(libs[0], q, k, sink, start, end, ws[0])
        b = run(libs[1], q, k, sink, start, end, ws[1])
        b2 = run(libs[1], q, k, sink, start, end, ws[1])
        torch.cuda.synchronize()
        d = (a.view(torch.int16) != b.view(torch.int16))
        d2 = (b.view(torch.int16) != b2.view(torch.int16))
        if d.any() or d2.any():
            bad += 1
            idx = d.nonzero()
            print(f"  shape {(H, Hkv, D, q_len, end - start)} iter {it}: A!=B at {int(d.sum())} scores, B!=B' at {int(d2.sum())};"
                  f" first {idx[:6].tolist()}  heads {sorted(set(idx[:, 1].tolist()))}  j range {int(idx[:, 2].min()) if len(idx) else -1}..{int(idx[:, 2].max()) if len(idx) else -1}")
    print(f"shape {(H, Hkv, D, sink, N, start, end, q_len)}: {bad}/{iters} iterations differ")
