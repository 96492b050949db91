#!/usr/bin/env python
"""How much does running the (independent) scoring calls of consecutive layers on TWO streams buy?  Tails of the persistent
kernels, launch gaps and the tiny merge / finalize kernels of one call overlap with the big kernels of the other."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import ops  # noqa: E402

dev = "cuda:0"
L, H, Hkv, D, sink, N, m = 28, 28, 4, 128, 32, 131072, 2000
q_len = m + 26
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(L, 1, H, q_len, D, generator=g, device=dev).half()
K = [torch.randn(1, Hkv, sink + 8192 + q_len, D, generator=g, device=dev).half() for _ in range(L)]
start = sink + 4000
outs = [torch.empty(1, Hkv, m, dtype=torch.float16, device=dev) for _ in range(L)]
need = 64 << 20
for nstreams in (1, 2, 3, 4, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    ws = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(6):
            for l in range(L):
                s = l % nstreams
                with torch.cuda.stream(streams[s]):
                    ops.score_chunk(Q[l], K[l], sink, start, start + m, out=outs[l], workspace=ws[s])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{nstreams} stream(s): {dt / (6 * L) * 1e6:.1f} us per score_chunk call")
