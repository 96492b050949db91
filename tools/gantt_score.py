#!/usr/bin/env python
"""Block-level schedule of persistent pass A from the `gantt` variant (KVZIP_HIP_LIB=tools/ab/lib_gantt.so)."""
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
buf = np.zeros(4 * 8192, dtype=np.uint64)
raw = C.CDLL(os.environ["KVZIP_HIP_LIB"])
raw.kvz_debug_read_trace.argtypes = [C.c_void_p, C.c_size_t]
assert raw.kvz_debug_read_trace(buf.ctypes.data, buf.nbytes) == 0
b = buf.reshape(-1, 4)
b = b[b[:, 0] > 0]
t0 = int(b[:, 0].min())
st = (b[:, 0].astype(np.int64) - t0) / 100.0  # us
en = (b[:, 1].astype(np.int64) - t0) / 100.0
items = (b[:, 3] >> 32).astype(np.int64)
tiles = (b[:, 3] & 0xffffffff).astype(np.int64)
hw = (b[:, 2] & 0xffffffff).astype(np.int64)
print(f"{len(b)} blocks; start min/max {st.min():.1f}/{st.max():.1f} us; end min/mean/max {en.min():.1f}/{en.mean():.1f}/{en.max():.1f} us")
print(f"items per block min/mean/max {items.min()}/{items.mean():.2f}/{items.max()}   tiles per block min/mean/max {tiles.min()}/{tiles.mean():.1f}/{tiles.max()}")
dur = en - st
print(f"us per tile: mean {np.mean(dur / np.maximum(tiles, 1)):.2f}  min {np.min(dur / np.maximum(tiles, 1)):.2f}  max {np.max(dur / np.maximum(tiles, 1)):.2f}")
mhz = b[:, 2].astype(np.float64) / np.maximum(dur, 1e-3)
print(f"shader clock over the block lifetime (s_memtime ticks / wall us): mean {mhz.mean():.0f} MHz  min {mhz.min():.0f}  max {mhz.max():.0f}")
late = np.argsort(st)[-10:]
print("latest-starting blocks (start,end,tiles):", [(round(float(st[i]), 1), round(float(en[i]), 1), int(tiles[i])) for i in late])
