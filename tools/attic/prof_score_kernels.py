#!/usr/bin/env python
"""Per-kernel hipEvent times of kvz_score_chunk at the bench geometry (uses the library's own profiler hook)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import _lib, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
H, Hkv, D, sink, N, m = 28, 4, 128, 32, 131072, 2000
q_len = m + 26
klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, H, q_len, D, generator=g, device=dev).half()
k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
start = sink + 60000
lib = _lib.load()
for _ in range(3):
    ops.score_chunk(q, k, sink, start, start + m)
torch.cuda.synchronize()
lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
for _ in range(iters):
    ops.score_chunk(q, k, sink, start, start + m)
torch.cuda.synchronize()
out = []
for name in ("score_rowstat", "score_colmax"):
    t, c = C.c_double(0), C.c_int64(0)
    lib.kvz_prof_read(name.encode(), C.byref(t), C.byref(c))
    out.append(f"{name} {t.value / max(c.value, 1) * 1e3:.1f} us")
print(os.environ.get("KVZIP_HIP_LIB", "default"), " | ".join(out))
