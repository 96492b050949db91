#!/usr/bin/env python
"""Pure host cost of one (update, _get_score) pair: tiny geometry (the kernels take a few us, the queue never fills), cProfile of
the Python path.  python tools/host_profile.py [n_streams] [unfused]"""
import cProfile, os, pstats, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd.kvcache import EvictCache  # noqa: E402
dev = "cuda:0"
L, H, Hkv, D, sink, N, m = 28, 28, 4, 128, 4, 4096, 32
q_len = m + 8
cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
g = torch.Generator(device=dev).manual_seed(0)
cap = sink + N + q_len + 64
store_k = [torch.randn(1, Hkv, cap, D, generator=g, device=dev).half() for _ in range(L)]
store_v = [torch.randn(1, Hkv, cap, D, generator=g, device=dev).half() for _ in range(L)]
Q = torch.randn(L, 1, H, q_len, D, generator=g, device=dev).half()
K = torch.randn(L, 1, Hkv, q_len, D, generator=g, device=dev).half()
kv = EvictCache(cfg, (sink, sink + N), device=dev, dtype=torch.float16, verbose=False)
kv.n_score_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kv.fuse_update_score = not (len(sys.argv) > 2 and sys.argv[2] == "unfused")   # update + _get_score = one library call
kv.adopt_dense(store_k, store_v, sink + N)
kv.init_score()
chunks = [(sink + c * m, sink + (c + 1) * m) for c in range(N // m)]
qs = [Q[l] for l in range(L)]; ks = [K[l] for l in range(L)]
def run(n_chunks):
    for st, en in chunks[:n_chunks]:
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        for l in range(L):
            k_all, _ = kv.update(ks[l], ks[l], l)
            kv._get_score(qs[l], k_all, l)
        kv.slice(seen)
run(4); kv._wait_score(); torch.cuda.synchronize()
kv.init_score()
t0 = time.perf_counter(); run(40); t1 = time.perf_counter(); kv._wait_score(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host {1e6 * (t1 - t0) / (40 * L):.1f} us per (update, _get_score) pair; GPU-complete {1e6 * (t2 - t0) / (40 * L):.1f} us")
kv.init_score()
pr = cProfile.Profile(); pr.enable(); run(40); pr.disable(); kv._wait_score(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
