import os, sys
sys.path.insert(0, os.getcwd())
import torch
from kvzip_amd import ops
dev = "cuda:0"; lib = ops._lib.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
g = torch.Generator(device=dev).manual_seed(0)
for (H, Hkv) in ((28, 4), (32, 8)):
    for q_len in (64, 128, 256, 512):
        for klen in (8192, 133000):
            if q_len > klen: continue
            D = 128; dt = torch.float16
            q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
            k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt); v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
            fl = 4.0 * H * D * (q_len * klen - q_len * (q_len - 1) / 2)
            res = {}
            for name, mb, sp in (("flash2+split", 1, 2), ("flash2", 1, 0), ("flash16", 1 << 30, 0)):
                p1 = lib.kvz_debug_set_tunable(b"flash2_min_blocks", mb); p2 = lib.kvz_debug_set_tunable(b"flash2_split", sp)
                try: res[name] = timeit(lambda: ops.flash_fwd(q, k, v))
                finally: lib.kvz_debug_set_tunable(b"flash2_min_blocks", p1); lib.kvz_debug_set_tunable(b"flash2_split", p2)
            units = ((q_len * (H // Hkv) + 255) // 256) * Hkv
            print(f"H {H} Hkv {Hkv} q {q_len:5d} k {klen:6d} units {units:4d}: " + " | ".join(f"{n} {us:8.1f} us {fl / us / 1e6:6.1f} TF" for n, us in res.items()), flush=True)
