#!/usr/bin/env python
"""Multi-row attention probe on the GPU box:
   (1) first-generation-step shape (4 heads x ~39k keys, G=7): time of the split-key decode kernel vs the multi-row kernel as
       q_len grows (where should kvz_varlen_attn switch?),
   (2) dense pre-prune forward shapes: kvz_flash_fwd vs torch SDPA, TFLOP/s."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = "cuda:0"

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

if len(sys.argv) > 1 and sys.argv[1] == "varlen":
    from kvzip_amd import ops
    D, G, Hkv = 128, 7, 4
    lens = [39000, 39500, 38800, 39900]
    g = torch.Generator(device=dev).manual_seed(0)
    starts, acc = [], 0
    for n in lens:
        starts.append(acc); acc += n + 1024
    k = torch.randn(acc, D, generator=g, device=dev).half(); v = torch.randn(acc, D, generator=g, device=dev).half()
    ks = torch.tensor(starts, dtype=torch.int32, device=dev); kl = torch.tensor(lens, dtype=torch.int32, device=dev)
    for q_len in (2, 4, 8, 10, 16, 19, 32, 64, 128, 512):
        q = torch.randn(Hkv * q_len, G, D, generator=g, device=dev).half()
        ws = ops.attn_workspace(Hkv, G, q_len, D, dev)
        us = timeit(lambda: ops.varlen_attn(q, k, v, ks, kl, q_len, max(lens), workspace=ws))
        print(f"q_len {q_len:4d} rows/head {q_len*G:5d}: {us:8.1f} us", flush=True)
    sys.exit(0)

for rows in ("1", "1000000"):
    print(f"--- KVZ_FLASH_MIN_ROWS={rows} ({'multi-row kernel' if rows == '1' else 'decode kernel'})", flush=True)
    subprocess.run([sys.executable, __file__, "varlen"], env={**os.environ, "KVZ_FLASH_MIN_ROWS": rows})

from kvzip_amd import ops
import torch.nn.functional as F
from torch.nn.attention.bias import causal_lower_right
g = torch.Generator(device=dev).manual_seed(0)
print("--- dense forward: kvz_flash_fwd vs torch SDPA", flush=True)
for (H, Hkv, q_len, klen) in ((28, 4, 2026, 2026), (28, 4, 16384, 16384), (28, 4, 2026, 35000), (28, 4, 2026, 133000), (32, 8, 2026, 35000)):
    D = 128
    q = torch.randn(1, H, q_len, D, generator=g, device=dev).half()
    k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half(); v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
    fl = 4.0 * H * D * (q_len * klen - (q_len * (q_len - 1) / 2 if True else 0))
    us = timeit(lambda: ops.flash_fwd(q, k, v), n=5)
    def sdpa():
        if q_len == klen:
            return F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
        return F.scaled_dot_product_attention(q, k, v, attn_mask=causal_lower_right(q_len, klen), enable_gqa=True)
    try:
        us2 = timeit(sdpa, n=5)
        a, b = ops.flash_fwd(q, k, v)[0].float(), sdpa().transpose(1, 2)[0].float()
        err = float((a - b).abs().max())
    except Exception as e:  # noqa: BLE001
        us2, err = float("nan"), str(e)[:60]
    print(f"H {H} Hkv {Hkv} q {q_len} k {klen}: flash_fwd {us:9.1f} us = {fl/us/1e6:6.1f} TFLOP/s | SDPA {us2:9.1f} us = {fl/us2/1e6:6.1f} TFLOP/s | max diff {err}", flush=True)
