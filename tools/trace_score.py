#!/usr/bin/env python
"""Read the in-kernel timeline written by the `trace` variant of tools/ablate_score.py (KVZIP_HIP_LIB=tools/ab/lib_trace.so).
Per tile: 9 stamps = [after-mfma-issue, after-epilogue] x 4 with one extra stamp after the turnover of the 4th block."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
H, Hkv, D, sink, N, m = 28, 4, 128, 32, 131072, 2000
q_len = m + 26
klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, H, q_len, D, generator=g, device=dev).half()
k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
start = sink + 60000
_lib.load()
for _ in range(3):
    ops.score_chunk(q, k, sink, start, start + m)
torch.cuda.synchronize()
buf = np.zeros(16 * 8 * 96, dtype=np.uint64)
raw = C.CDLL(os.environ["KVZIP_HIP_LIB"])
raw.kvz_debug_read_trace.argtypes = [C.c_void_p, C.c_size_t]
assert raw.kvz_debug_read_trace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(16, 8, 32, 3).astype(np.int64)
for x in range(2):
    n = int((tr[x, 0, :, 0] > 0).sum())
    if n < 3:
        continue
    t0 = tr[x, :, 0, 0].min()
    print(f"block {x * 16 + 5}: {n} tiles traced; per tile: start (relative to block start), then per wave [compute-until-barrier / wait-at-barrier]")
    for ti in range(min(n, 14)):
        start = tr[x, :, ti, 0] - t0
        comp = tr[x, :, ti, 1] - tr[x, :, ti, 0]
        wait = tr[x, :, ti, 2] - tr[x, :, ti, 1]
        nxt = (tr[x, :, ti + 1, 0] - tr[x, :, ti, 2]) if ti + 1 < n else np.zeros(8, dtype=np.int64)
        print(f"  tile {ti:2d} start {int(start.min()):7d}  " + " ".join(f"{int(c)}/{int(w)}" for c, w in zip(comp, wait)) + f"   after-barrier part (epilogue etc.): {int(nxt.mean())}")
