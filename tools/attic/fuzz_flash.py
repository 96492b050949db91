#!/usr/bin/env python
"""Fuzz the dense forward: random (heads, KV heads, query positions, keys, dtype) with the 32-row kernel forced, the last round of blocks
split along the keys (flash2_split = 2) against one block per unit (0) and against an fp32 reference with the bottom-right causal mask.
   python tools/fuzz_flash.py [n_shapes] [seed]"""
import math, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import ops
dev = "cuda:0"
lib = ops._lib.load()
n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator(device=dev).manual_seed(1)
worst = {"split_vs_ref": 0.0, "unit_vs_ref": 0.0, "split_vs_unit": 0.0, "lse": 0.0}
fails = 0
prev_b = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
try:
    for it in range(n_shapes):
        Hkv = rng.choice([1, 2, 3, 4, 8, 16])
        G = rng.choice([1, 2, 4, 5, 7])
        H = Hkv * G
        q_len = rng.choice([1, 7, 33, 64, 100, 257, 300, 513, 700, 1025])
        klen = max(1, q_len + rng.choice([-q_len // 2, 0, 1, 63, 64, 65, 500, 2000, 5000]))
        dt = rng.choice([torch.float16, torch.bfloat16])
        D = 128
        q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
        k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
        v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
        outs = {}
        for name, sp in (("split", 2), ("unit", 0)):
            p = lib.kvz_debug_set_tunable(b"flash2_split", sp)
            try:
                outs[name] = ops.flash_fwd(q, k, v, causal=True, return_lse=True)
            finally:
                lib.kvz_debug_set_tunable(b"flash2_split", p)
        s = torch.einsum("hid,hjd->hij", q[0].float(), k[0].float().repeat_interleave(G, 0)) / math.sqrt(D)
        i = torch.arange(q_len, device=dev).view(1, q_len, 1)
        j = torch.arange(klen, device=dev).view(1, 1, klen)
        s = s.masked_fill(j > i + (klen - q_len), float("-inf"))
        want = torch.einsum("hij,hjd->ihd", torch.nan_to_num(torch.softmax(s, -1), nan=0.0), v[0].float().repeat_interleave(G, 0))
        lse_ref = torch.logsumexp(s, -1)
        tol = 1e-3 if dt == torch.float16 else 8e-3
        step = torch.pow(2.0, torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -14))) - (10 if dt == torch.float16 else 7))
        bound = torch.maximum(torch.full_like(want, tol), step * 1.001)
        e_s = (outs["split"][0][0].float() - want.to(dt).float()).abs()
        e_u = (outs["unit"][0][0].float() - want.to(dt).float()).abs()
        e_su = (outs["split"][0][0].float() - outs["unit"][0][0].float()).abs()
        fin = torch.isfinite(lse_ref)
        e_l = (outs["split"][1][0][fin] - lse_ref[fin]).abs().max().item() if fin.any() else 0.0
        ok = bool((e_s <= bound).all()) and bool((e_u <= bound).all()) and e_l <= 1e-3 and bool(torch.isinf(outs["split"][1][0][~fin]).all())
        worst["split_vs_ref"] = max(worst["split_vs_ref"], float((e_s / bound).max()))
        worst["unit_vs_ref"] = max(worst["unit_vs_ref"], float((e_u / bound).max()))
        worst["split_vs_unit"] = max(worst["split_vs_unit"], float((e_su / bound).max()))
        worst["lse"] = max(worst["lse"], e_l)
        if not ok:
            fails += 1
            print(f"FAIL H {H} Hkv {Hkv} q {q_len} k {klen} {dt}: split {float(e_s.max()):.2e} unit {float(e_u.max()):.2e} lse {e_l:.2e}", flush=True)
finally:
    lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev_b)
print(f"{n_shapes} shapes: {fails} failures; worst error / bound: split vs fp32 reference {worst['split_vs_ref']:.3f}, one block per unit vs reference "
      f"{worst['unit_vs_ref']:.3f}, split vs one block per unit {worst['split_vs_unit']:.3f}; worst LSE error {worst['lse']:.2e}")
