#!/usr/bin/env python
"""Debug: per-block timeline of score_rowstat (needs a -DKVZ_TRACE build passed via KVZIP_HIP_LIB)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops
dev = "cuda:0"
H, Hkv, D, sink, N, m = 28, 4, 128, 32, 131072, 2000
q_len = m + 26
klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, H, q_len, D, generator=g, device=dev).half()
k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
start = sink + 60000
lib = C.CDLL(_lib.LIB_PATH)
nblk = 4096
trace = torch.zeros(nblk * 6, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.score_chunk(q, k, sink, start, start + m)
torch.cuda.synchronize()
lib.kvz_debug_set_trace(C.c_void_p(trace.data_ptr()))
ops.score_chunk(q, k, sink, start, start + m)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nblk, 6)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
start_, first, end_, done, hw, ntile = [t[:, i] for i in range(6)]
print("blocks traced", len(t), "wall_clock ticks are 100 MHz (10 ns)")
print("kernel span (us):", (done.max() - t0) / 100.0)
print("block start (us) percentiles:", np.percentile((start_ - t0) / 100.0, [0, 25, 50, 75, 100]).round(1))
print("block duration (us) pct:", np.percentile((done - start_) / 100.0, [0, 25, 50, 75, 100]).round(1))
ok = ntile > 0
print("prologue (us) pct:", np.percentile((first[ok] - start_[ok]) / 100.0, [0, 25, 50, 75, 100]).round(1))
print("per-tile (us) pct:", np.percentile((end_[ok] - first[ok]) / 100.0 / ntile[ok], [0, 25, 50, 75, 100]).round(2))
print("tiles per block:", np.bincount(ntile.astype(int)))
cu = (hw & 0xFFFFFFFF)
xcc = hw >> 32
print("distinct (xcc, hwid>>8 cu bits):", len(set(zip(xcc.tolist(), ((cu >> 8) & 0xFFFF).tolist()))))
order = np.argsort(start_)
print("first 12 blocks by start: start/dur us, tiles:", [(round((start_[i]-t0)/100.,1), round((done[i]-start_[i])/100.,1), int(ntile[i])) for i in order[:12]])
print("last 12 blocks by start:", [(round((start_[i]-t0)/100.,1), round((done[i]-start_[i])/100.,1), int(ntile[i])) for i in order[-12:]])
