#!/usr/bin/env python
"""How do the scoring kernels scale with the amount of work?  (G = query heads per KV head, Hkv, m)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops
lib = _lib.load()
dev = "cuda:0"
def run(Hkv, G, m, D=128, sink=32, N=131072, iters=10):
    q_len = m + 26
    klen = sink + N + q_len
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).half()
    k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
    start = sink + 60000
    for _ in range(3): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
    for _ in range(iters): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_enable(0)
    out = []
    for name in ("score_rowstat", "score_colmax"):
        t, c = C.c_double(0), C.c_int64(0)
        lib.kvz_prof_read(name.encode(), C.byref(t), C.byref(c))
        out.append(t.value / max(c.value, 1) * 1e3)
    logits_a = Hkv * G * q_len * (sink + m + q_len / 2)
    logits_b = Hkv * G * q_len * m
    print(f"Hkv={Hkv} G={G} m={m}: rowstat {out[0]:7.1f} us ({logits_a / out[0] / 1e3:6.0f} Mlogit/ms)   colmax {out[1]:7.1f} us ({logits_b / out[1] / 1e3:6.0f} Mlogit/ms)")
for Hkv, G, m in ((4, 7, 2000), (4, 4, 2000), (4, 2, 2000), (4, 1, 2000), (8, 7, 2000), (16, 7, 2000), (2, 7, 2000), (1, 7, 2000), (4, 7, 1000), (4, 7, 500)):
    run(Hkv, G, m)
