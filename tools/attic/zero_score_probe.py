#!/usr/bin/env python
"""Clock / power sensitivity of the scoring kernels: the same launches on random, small-magnitude and all-zero inputs
(identical instruction streams; only the data toggling, hence the power and the DVFS clock, differ)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops
lib = _lib.load(); dev = "cuda:0"
Hkv, G, m, D, sink, N = 4, 7, 2000, 128, 32, 131072
q_len = m + 26; klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
q0 = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev); k0 = torch.randn(1, Hkv, klen, D, generator=g, device=dev)
start = sink + 60000
def run(tag, q, k):
    for _ in range(5): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
    for _ in range(40): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_enable(0)
    r = []
    for kn in ("score_rowstat", "score_colmax"):
        t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(kn.encode(), C.byref(t), C.byref(c))
        r.append(round(t.value / max(c.value, 1) * 1e3, 1))
    print(f"{tag:34s} rowstat {r[0]:6.1f} us   colmax {r[1]:6.1f} us", flush=True)
for rnd in range(2):
    run("random N(0,1)", q0.half(), k0.half())
    run("all zero", torch.zeros_like(q0).half(), torch.zeros_like(k0).half())
    run("constant 0.125", torch.full_like(q0, 0.125).half(), torch.full_like(k0, 0.125).half())
    run("random, q only (k zero)", q0.half(), torch.zeros_like(k0).half())
    run("random sign, |x| = 1", torch.sign(q0).half(), torch.sign(k0).half())
