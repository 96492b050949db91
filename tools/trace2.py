#!/usr/bin/env python
"""Read the in-kernel timeline of the pipelined row-statistics kernel (library built with -DKVZ_TRACE=1):
   KVZIP_HIP_LIB=tools/ab/lib_trace.so python tools/trace2.py
Per tile and wave: durations of the four steps (step 2 split into before-barrier / barrier wait / hand-over / rest), gap to the next tile."""
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
buf = np.zeros(8 * 8 * 40 * 16, dtype=np.uint64)
raw = C.CDLL(os.environ["KVZIP_HIP_LIB"]); raw.kvz_debug_read_trace2.argtypes = [C.c_void_p, C.c_size_t]
assert raw.kvz_debug_read_trace2(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(8, 8, 40, 16).astype(np.int64)
for x in range(2):
    n = int((tr[x, 0, :, 0] > 0).sum())
    if n < 3: continue
    t0 = tr[x, :, 0, 0].min()
    print(f"block {x * 32 + 5}: {n} tiles; per wave: step0 step1 | step2: pre-barrier, wait, hand-over, rest | step3 | gap to next tile   (s_memtime ticks)")
    for ti in range(min(n, 26)):
        r = tr[x, :, ti]
        nxt = tr[x, :, ti + 1, 0] if ti + 1 < n else r[:, 7]
        print(f" tile {ti:2d} start {int(r[:, 0].min() - t0):7d}")
        for w in range(8):
            a = r[w]
            sw = ""
            if ti > 0 and a[15] > tr[x, w, ti - 1, 7] and a[15] <= a[0]:  # an item switch happened between the previous tile and this one
                p7 = tr[x, w, ti - 1, 7]
                sw = f"   switch: dump {a[8]-p7} stats {a[9]-a[8]} read_q {a[10]-a[9]} chain0 {a[11]-a[10]} item_from {a[12]-a[11]} rows_of {a[13]-a[12]} stage_q {a[14]-a[13]} start_item {a[15]-a[14]} -> tile {a[0]-a[15]}"
            print(f"    w{w}: {a[1]-a[0]:5d} {a[2]-a[1]:5d} | {a[4]-a[2]:5d} {a[5]-a[4]:5d} {a[6]-a[5]:5d} {a[3]-a[6]:5d} | {a[7]-a[3]:5d} | {nxt[w]-a[7]:6d}   (start +{a[0]-r[:,0].min():5d})" + sw)
