#!/usr/bin/env python
"""Decode attention over a whole model's worth of pruned caches (28 layers x 80 MB: every call streams COLD HBM, unlike
tools/attn_probe.py whose single 80-MB cache stays in the 256-MB infinity cache): per-token time of the loop over the layers for
several values of the `attn_items` knob, fused append on / off.   python tools/decode_cold_probe.py [items ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"
L, Hkv, G, D = 28, 4, 7, 128
lens = [39000, 39500, 38800, 39900]
slack = 1024
g = torch.Generator(device=dev).manual_seed(0)
starts, acc = [], 0
for n in lens:
    starts.append(acc); acc += n + slack
K = [torch.randn(acc, D, generator=g, device=dev).half() for _ in range(L)]
V = [torch.randn(acc, D, generator=g, device=dev).half() for _ in range(L)]
q = [torch.randn(Hkv, G, D, generator=g, device=dev).half() for _ in range(L)]
kn = [torch.randn(1, Hkv, 1, D, generator=g, device=dev).half() for _ in range(L)]
ks = torch.tensor(starts, dtype=torch.int32, device=dev); kl = torch.tensor(lens, dtype=torch.int32, device=dev)
meta = ops._meta_host(starts, lens, Hkv)
byts = 2 * sum(lens) * D * 2
outs = [torch.empty(Hkv, G, D, dtype=torch.float16, device=dev) for _ in range(L)]

def token(ws, fused, off):
    for l in range(L):
        if fused:
            ops.varlen_attn_append(q[l], K[l], V[l], kn[l], kn[l], ks, kl, off, max(lens) + off + 1, workspace=ws, meta_host=meta, out=outs[l])
        else:
            ops.varlen_attn(q[l], K[l], V[l], ks, kl, 1, max(lens), workspace=ws, meta_host=meta, out=outs[l])

items = [int(x) for x in sys.argv[1:]] or [128, 176, 192, 224, 256, 320, 384, 512]
ref = None
for it in items:
    prev = lib.kvz_debug_set_tunable(b"attn_items", it)
    ws = ops.attn_workspace(Hkv, G, 1, D, dev)
    res = {}
    for fused in (False, True):
        for _ in range(2): token(ws, fused, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for t in range(20): token(ws, fused, t % 8)
        e1.record(); torch.cuda.synchronize()
        res[fused] = e0.elapsed_time(e1) / 20
    token(ws, False, 0); torch.cuda.synchronize()
    o = torch.stack(outs).float()
    if ref is None: ref = o
    lib.kvz_debug_set_tunable(b"attn_items", prev)
    print(f"attn_items {it:4d}: plain {res[False]:.3f} ms/token = {res[False] / L * 1e3:5.1f} us/layer = {byts / (res[False] / L * 1e-3) / 1e12:4.2f} TB/s | "
          f"fused append {res[True]:.3f} ms/token = {res[True] / L * 1e3:5.1f} us/layer | max |diff to first| {float((o - ref).abs().max()):.1e}", flush=True)
