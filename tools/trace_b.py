#!/usr/bin/env python
"""Read the in-kernel timeline of the column-maximum kernel (pass B; library built with -DKVZ_TRACE=1):
   KVZIP_HIP_LIB=tools/ab/lib_trace.so python tools/trace_b.py
Per tile and wave: the four steps (step 2 split into before the hand-over / DMA issue + counted wait / barrier wait / rest), plus the
block's prologue (entry -> first tile) and epilogue (last tile -> exit).  s_memtime ticks."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops
dev = "cuda:0"
H, Hkv, D, sink, N, m = 28, 4, 128, 32, 131072, 2000
q_len = m + 26; klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, H, q_len, D, generator=g, device=dev).half(); k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
start = sink + 60000
_lib.load()
for _ in range(3): ops.score_chunk(q, k, sink, start, start + m)
torch.cuda.synchronize()
buf = np.zeros(8 * 8 * 20 * 8, dtype=np.uint64)
raw = C.CDLL(os.environ["KVZIP_HIP_LIB"]); raw.kvz_debug_read_trace_b.argtypes = [C.c_void_p, C.c_size_t]
assert raw.kvz_debug_read_trace_b(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(8, 8, 20, 8).astype(np.int64)
for x in range(3):
    n = int((tr[x, 0, :19, 0] > 0).sum())
    if n < 2: continue
    ent = tr[x, :, 19]
    t0 = ent[:, 0].min()
    print(f"block {x * 32 + 5}: {n} tiles.  prologue (entry -> tile loop) per wave: {[int(v) for v in ent[:, 1] - ent[:, 0]]}; "
          f"epilogue (last tile -> exit): {[int(v) for v in ent[:, 3] - ent[:, 2]]}; whole block {int(ent[:, 3].max() - t0)} ticks")
    print("   per wave: step0 step1 | step2: pre-hand-over, DMA issue + counted wait, barrier wait, rest | step3 | (tile start offset)")
    for ti in range(n):
        r = tr[x, :, ti]
        print(f" tile {ti:2d} start {int(r[:, 0].min() - t0):7d}  tile time (slowest wave) {int((r[:, 7] - r[:, 0]).max()):6d}")
        for w in range(8):
            a = r[w]
            print(f"    w{w}: {a[1]-a[0]:5d} {a[2]-a[1]:5d} | {a[4]-a[2]:5d} {a[5]-a[4]:5d} {a[6]-a[5]:5d} {a[3]-a[6]:5d} | {a[7]-a[3]:5d}   (+{a[0]-r[:,0].min():5d})")
