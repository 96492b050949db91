#!/usr/bin/env python
"""Host-side cost of the scoring loop: time to ENQUEUE 10 chunks x 28 layers of (update, _get_score) versus the GPU time."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd.kvcache import EvictCache  # noqa: E402

dev = "cuda:0"
L, H, Hkv, D, sink, N, m = 28, 28, 4, 128, 32, 131072, 2000
q_len = m + 26
cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
g = torch.Generator(device=dev).manual_seed(0)
cap = sink + N + q_len + 64
store_k = [torch.randn(1, Hkv, cap, D, generator=g, device=dev).half() for _ in range(L)]
store_v = [torch.randn(1, Hkv, cap, D, generator=g, device=dev).half() for _ in range(L)]
Q = torch.randn(L, 1, H, q_len, D, generator=g, device=dev).half()
K = torch.randn(L, 1, Hkv, q_len, D, generator=g, device=dev).half()
V = torch.randn(L, 1, Hkv, q_len, D, generator=g, device=dev).half()
MODE = sys.argv[2] if len(sys.argv) > 2 else "plain"
kv = EvictCache(cfg, (sink, sink + N), device=dev, dtype=torch.float16, verbose=False)
kv.n_score_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kv.adopt_dense(store_k, store_v, sink + N)
kv.init_score()
chunks = [(sink + c * m, sink + (c + 1) * m) for c in range(32)]
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st, en in chunks[rep * 10: rep * 10 + 10]:
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        for l in range(L):
            if MODE == "views":
                k_all, _ = kv.update(K[l][:, :, :q_len], V[l][:, :, :q_len], l)
                kv._get_score(Q[l][:, :, :q_len], k_all, l)
            elif MODE == "kv":
                k_all, _ = kv.update(K[l], V[l], l)
                kv._get_score(Q[l], k_all, l)
            else:
                k_all, _ = kv.update(K[l], K[l], l)
                kv._get_score(Q[l], k_all, l)
        kv.slice(seen)
    kv._wait_score()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = 10 * L
    print(f"{MODE} streams {kv.n_score_streams} rep {rep}: enqueue {1e6 * (t1 - t0) / n:.1f} us per (layer, chunk); GPU-complete {1e6 * (t2 - t0) / n:.1f} us per (layer, chunk)")
