#!/usr/bin/env python
"""hipEvent time of the all-layer compaction gather at the bench geometry."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops
lib = _lib.load()
dev = "cuda:0"
L, Hkv, D, sink, N = 28, 4, 128, 32, 131072
g = torch.Generator(device=dev).manual_seed(0)
ks = [torch.randn(1, Hkv, sink + N, D, generator=g, device=dev).half() for _ in range(L)]
vs = [torch.randn(1, Hkv, sink + N, D, generator=g, device=dev).half() for _ in range(L)]
valid = torch.rand(L, 1, Hkv, N, generator=g, device=dev) < 0.3
plan = ops.compact_plan(valid, sink, sink + N, slack=1024)
totals = (plan.len_k.cpu().sum(-1) + 1024 * Hkv).tolist()
for _ in range(2):
    ko, vo = ops.compact_layers(ks, vs, plan, totals)
torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
for _ in range(10):
    ko, vo = ops.compact_layers(ks, vs, plan, totals)
torch.cuda.synchronize(); lib.kvz_prof_enable(0)
t, c = C.c_double(0), C.c_int64(0)
lib.kvz_prof_read(b"compact_gather", C.byref(t), C.byref(c))
kept = int(plan.len_k.sum())
byts = 2 * 2 * kept * D * 2 + L * Hkv * N
ms = t.value / c.value
print(f"compact_gather {ms * 1e3:.1f} us, {byts / ms / 1e6:.0f} GB/s ({byts / ms / 1e6 / 8000:.3f} of 8 TB/s)")
