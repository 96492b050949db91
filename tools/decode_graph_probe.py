#!/usr/bin/env python
"""Why does the replayed HIP graph of a generation step not beat the step issued layer by layer from Python?  A pruned 128k cache
(Qwen2.5-7B geometry, ratio 0.3), MODE=loop / graph: 24 tokens each.  Run under rocprofv3 --kernel-trace and summarise with
tools/decode_trace_summary.py (kernel durations and the gaps between consecutive kernels of a token)."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd.kvcache import EvictCache  # noqa: E402

mode = os.environ.get("MODE", "loop")
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
T = 24
if mode == "graph":
    dg = kv.decode_graph(q, k, k)
    for _ in range(3):
        dg.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(T):
        dg.replay()
    torch.cuda.synchronize()
else:
    for _ in range(3):
        for l in range(L):
            kv.update_attend(q[l], k[l], k[l], l)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(T):
        for l in range(L):
            kv.update_attend(q[l], k[l], k[l], l)
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{mode}: {dt / T * 1e3:.3f} ms per token, {dt / T / L * 1e6:.2f} us per layer")
