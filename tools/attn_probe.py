#!/usr/bin/env python
"""Decode-attention probe on the GPU box: time per call (hipEvents around a loop) and effective HBM bandwidth for
   * the headline cache (4 heads of ~39 k keys), * AdaKV-style ragged heads [39k, 4k, 120k, 500], * a head-level layer (3 of 8 heads kept),
   with and without the head segments as kernel arguments; parity vs the CPU oracle on the ragged case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ctypes as C
from kvzip_amd import ops, _lib
import kvzip_oracle as orc
lib = _lib.load()
dev = "cuda:0"
D, G = 128, 7
g = torch.Generator(device=dev).manual_seed(0)
def case(name, lens, G=7, dt=torch.float16, check=False):
    Hkv = len(lens)
    slack = 1024
    starts, acc = [], 0
    for n in lens:
        starts.append(acc); acc += n + slack
    k = torch.randn(acc, D, generator=g, device=dev).to(dt); v = torch.randn(acc, D, generator=g, device=dev).to(dt)
    q = torch.randn(Hkv, G, D, generator=g, device=dev).to(dt)
    ks = torch.tensor(starts, dtype=torch.int32, device=dev); kl = torch.tensor(lens, dtype=torch.int32, device=dev)
    ws = ops.attn_workspace(Hkv, G, 1, D, dev)
    meta = ops._meta_host(starts, lens, Hkv)
    out = {}
    for tag, m in (("args", meta), ("dev", None)):
        for _ in range(5): ops.varlen_attn(q, k, v, ks, kl, 1, max(lens), workspace=ws, meta_host=m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): ops.varlen_attn(q, k, v, ks, kl, 1, max(lens), workspace=ws, meta_host=m)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        out[tag] = us
        lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
        for _ in range(50): ops.varlen_attn(q, k, v, ks, kl, 1, max(lens), workspace=ws, meta_host=m)
        torch.cuda.synchronize(); lib.kvz_prof_enable(0)
        t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(b"varlen_attn", C.byref(t), C.byref(c))
        out[tag + "_gpu"] = t.value / max(c.value, 1) * 1e3
    byts = 2 * sum(lens) * D * 2
    err = None
    if check:
        got = ops.varlen_attn(q, k, v, ks, kl, 1, max(lens), workspace=ws, meta_host=meta).cpu().float()
        want = orc.varlen_attn(q.cpu(), k.cpu(), v.cpu(), starts, lens, 1).float()
        err = float((got - want).abs().max())
    print(f"{name:34s} {byts/1e6:7.1f} MB  loop: args {out['args']:6.1f} us  dev {out['dev']:6.1f} us | bracketed GPU time: args {out['args_gpu']:6.1f} us = {byts/out['args_gpu']/1e6:5.2f} TB/s   dev {out['dev_gpu']:6.1f} us = {byts/out['dev_gpu']/1e6:5.2f} TB/s"
          + (f"   max err vs oracle {err:.2e}" if err is not None else ""), flush=True)
case("headline 4 x ~39k", [39000, 39500, 38800, 39900])
case("ragged [39k,4k,120k,500]", [39000, 4000, 120000, 500], check=True)
case("head-level 3 of 8 kept (131k)", [131104, 32, 32, 131104, 32, 32, 131104, 32], G=5)
case("llama 8 x ~39k (G=4)", [39000] * 8, G=4)
case("short 4 x 2k", [2000, 2100, 1900, 2050])
