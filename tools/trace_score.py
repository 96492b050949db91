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
buf = np.zeros(16 * 4 * 192, dtype=np.uint64)
raw = C.CDLL(os.environ["KVZIP_HIP_LIB"])
raw.kvz_debug_read_trace.argtypes = [C.c_void_p, C.c_size_t]
assert raw.kvz_debug_read_trace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(16, 4, 192)
for x in range(3):
    for w in range(4):
        s = tr[x, w]
        st = [int(v) for v in s[2:] if v]
        if len(st) < 14:
            continue
        d = np.diff(st)
        print(f"block {x * 16 + 5} wave {w}: {len(st)} stamps, first gap (startup) {d[0]}")
        if w in (0, 2):
            body = d[1:1 + ((len(d) - 1) // 11) * 11].reshape(-1, 11)
            for ti, r in enumerate(body[:12]):
                print(f"    tile {ti:2d}  mfma/epi: {r[0]}/{r[1]} {r[2]}/{r[3]} {r[4]}/{r[5]}  last: mfma {r[6]} dma-wait {r[7]} barrier {r[8]} refill+frags {r[9]} epi {r[10]}   sum {r.sum()}")
