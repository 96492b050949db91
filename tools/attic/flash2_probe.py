#!/usr/bin/env python
"""Dense forward shapes: the 32-row kernel (kvz_flash2.hip) vs the 16-row kernel (kvz_flash.hip) vs torch SDPA, TFLOP/s and max
   difference, fp16 and bf16.  python tools/flash2_probe.py [--no-sdpa]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from torch.nn.attention.bias import causal_lower_right
from kvzip_amd import ops
dev = "cuda:0"
lib = ops._lib.load()

def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator(device=dev).manual_seed(0)
shapes = ((28, 4, 2026, 2026), (28, 4, 16000, 16000), (28, 4, 2026, 35000), (28, 4, 2026, 133000), (32, 8, 2026, 133000), (40, 8, 2026, 35000))
for dt in (torch.float16, torch.bfloat16):
    for (H, Hkv, q_len, klen) in shapes:
        D = 128
        q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
        k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt); v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
        fl = 4.0 * H * D * (q_len * klen - q_len * (q_len - 1) / 2)
        res = {}
        for name, mb in (("flash2", 1), ("flash16", 1 << 30)):
            prev = lib.kvz_debug_set_tunable(b"flash2_min_blocks", mb)
            try:
                res[name] = (timeit(lambda: ops.flash_fwd(q, k, v)), ops.flash_fwd(q, k, v)[0].float())
            finally:
                lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev)
        line = f"{str(dt)[6:]:8s} H {H} Hkv {Hkv} q {q_len} k {klen}: " + " | ".join(f"{n} {us:9.1f} us = {fl / us / 1e6:6.1f} TFLOP/s" for n, (us, _) in res.items())
        line += f" | max |flash2 - flash16| {float((res['flash2'][1] - res['flash16'][1]).abs().max()):.2e}"
        if "--no-sdpa" not in sys.argv and q_len * klen <= 2026 * 35000:
            def sdpa():
                if q_len == klen:
                    return F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
                return F.scaled_dot_product_attention(q, k, v, attn_mask=causal_lower_right(q_len, klen), enable_gqa=True)
            us2 = timeit(sdpa, n=3)
            line += f" | SDPA {us2:9.1f} us = {fl / us2 / 1e6:6.1f} TFLOP/s, max |flash2 - SDPA| {float((res['flash2'][1] - sdpa().transpose(1, 2)[0].float()).abs().max()):.2e}"
        print(line, flush=True)

# ---- f2: the same forward with the scoring window's statistics switched on (kvz_flash_fwd_window) ----
print("--- forward with / without the window statistics (sink 32, chunk of 2000 keys, repeat chunk = the query rows)", flush=True)
prev = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
try:
    for dt in (torch.float16, torch.bfloat16):
        for (H, Hkv, q_len, klen) in ((28, 4, 2026, 35000), (28, 4, 2026, 133000)):
            D = 128
            q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
            k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt); v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
            stats = torch.empty(Hkv, (H // Hkv * q_len + 127) // 128 * 128, 2, dtype=torch.float32, device=dev)
            start = 32 + 10000
            us0 = timeit(lambda: ops.flash_fwd(q, k, v))
            us1 = timeit(lambda: ops.flash_fwd_window(q, k, v, 32, start, start + 2000, stats))
            same = torch.equal(ops.flash_fwd(q, k, v), ops.flash_fwd_window(q, k, v, 32, start, start + 2000, stats))
            print(f"{str(dt)[6:]:8s} q {q_len} k {klen}: plain {us0:8.1f} us | with statistics {us1:8.1f} us (+{(us1 / us0 - 1) * 100:.1f} %) | outputs identical: {same}", flush=True)
finally:
    lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev)
