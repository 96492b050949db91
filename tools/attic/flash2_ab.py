#!/usr/bin/env python
"""One library (KVZIP_HIP_LIB) on the dense-forward shapes: TFLOP/s and a checksum of the output, for same-box A/B runs of kernel
variants:   for lib in a.so b.so; do KVZIP_HIP_LIB=$PWD/$lib python tools/flash2_ab.py; done"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import ops
dev = "cuda:0"
lib = ops._lib.load()


def timeit(fn, n=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


g = torch.Generator(device=dev).manual_seed(0)
shapes = ((torch.float16, 28, 4, 2026, 133000), (torch.float16, 28, 4, 16000, 16000), (torch.bfloat16, 28, 4, 2026, 133000), (torch.float16, 32, 8, 2026, 35000),
          (torch.float16, 40, 8, 2026, 133000), (torch.float16, 40, 8, 2026, 35000), (torch.float16, 24, 8, 2026, 35000))
prev_b = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
tag = os.path.basename(os.environ.get("KVZIP_HIP_LIB", "default"))
for kv in filter(None, os.environ.get("KVZ_TUNE", "").split(",")):   # e.g. KVZ_TUNE=flash_rows64=1,flash2_split=2
    k_, v_ = kv.split("=")
    lib.kvz_debug_set_tunable(k_.encode(), int(v_))
    tag += " " + kv
if "F2_SPLIT" in os.environ:   # (libraries that know the knob: 0 = one block per unit)
    lib.kvz_debug_set_tunable(b"flash2_split", int(os.environ["F2_SPLIT"]))
    tag += " split=" + os.environ["F2_SPLIT"]
for (dt, H, Hkv, q_len, klen) in shapes:
    D = 128
    q = torch.randn(1, H, q_len, D, generator=g, device=dev).to(dt)
    k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt); v = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
    fl = 4.0 * H * D * (q_len * klen - q_len * (q_len - 1) / 2)
    us = timeit(lambda: ops.flash_fwd(q, k, v))
    out = ops.flash_fwd(q, k, v)
    out = out[0] if isinstance(out, (tuple, list)) else out
    h = hashlib.sha1(out.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"{tag:24s} {str(dt)[6:]:8s} H {H} Hkv {Hkv} q {q_len} k {klen}: {us:9.1f} us = {fl / us / 1e6:7.1f} TFLOP/s  out sha1 {h}", flush=True)
lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev_b)
