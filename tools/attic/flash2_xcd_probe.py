#!/usr/bin/env python
"""The 32-row dense forward with the XCD-aware block order (round 4) vs the head-major grid of round 3, same process, interleaved:
   python tools/flash2_xcd_probe.py   ->  TFLOP/s per shape and order, outputs bit-identical or not."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import ops
dev = "cuda:0"
lib = ops._lib.load()

def timeit(fn, n=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator(device=dev).manual_seed(0)
shapes = ((28, 4, 2026, 133000), (28, 4, 2026, 35000), (28, 4, 16000, 16000), (32, 8, 2026, 133000), (40, 8, 2026, 133000), (40, 8, 2026, 35000))
prev_b = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
for dt in (torch.float16, torch.bfloat16):
    for (H, Hkv, q_len, klen) in shapes:
        D = 128
        q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
        k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt); v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
        fl = 4.0 * H * D * (q_len * klen - q_len * (q_len - 1) / 2)
        res, outs = {"xcd": [], "plain": []}, {}
        for rnd in range(3):
            for name, x in (("xcd", 1), ("plain", 0)):
                lib.kvz_debug_set_tunable(b"flash2_xcd", x)
                res[name].append(timeit(lambda: ops.flash_fwd(q, k, v)))
                outs[name] = ops.flash_fwd(q, k, v)
        lib.kvz_debug_set_tunable(b"flash2_xcd", -1)
        print(f"{str(dt)[6:]:8s} H {H} Hkv {Hkv} q {q_len} k {klen}: " + " | ".join(
            f"{n} {min(us):9.1f} us = {fl / min(us) / 1e6:6.1f} TFLOP/s (runs {[round(u) for u in us]})" for n, us in res.items())
            + f" | identical: {torch.equal(outs['xcd'], outs['plain'])}", flush=True)
lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev_b)
