#!/usr/bin/env python
"""Decode step timing on a pruned 128k cache (Qwen2.5-7B geometry, ratio 0.3): per-layer time of update_attend."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd.kvcache import EvictCache  # noqa: E402

dev = "cuda:0"
L, H, Hkv, D, sink, N = 28, 28, 4, 128, 32, 131072
cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
g = torch.Generator(device=dev).manual_seed(0)
kv = EvictCache(cfg, (sink, sink + N), device=dev, dtype=torch.float16, verbose=False)
for l in range(L):
    kv.update(torch.randn(1, Hkv, sink + N, D, generator=g, device=dev).half(), torch.randn(1, Hkv, sink + N, D, generator=g, device=dev).half(), l)
kv.score = [torch.rand(1, Hkv, N, generator=g, device=dev).half() for _ in range(L)]
kv.prune(0.3)
q = torch.randn(L, 1, H, 1, D, generator=g, device=dev).half()
k = torch.randn(L, 1, Hkv, 1, D, generator=g, device=dev).half()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        for l in range(L):
            kv.update_attend(q[l], k[l], k[l], l)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"KVZ_ATTN_ITEMS={os.environ.get('KVZ_ATTN_ITEMS', '512')}: {dt / 16 * 1e3:.3f} ms per token, {dt / 16 / L * 1e6:.1f} us per layer")
