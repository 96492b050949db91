#!/usr/bin/env python
"""How much does running the (independent) scoring calls of consecutive layers on TWO streams buy, and does it depend on
how the side streams were created?  HIP maps streams onto a few hardware queues; two streams on one queue do not overlap.
   python tools/overlap_probe.py [n_dummy_streams_created_first]"""
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
dummies = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]


def run(streams, label):
    ws = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in streams]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(6):
            for l in range(L):
                s = l % len(streams)
                ops.score_chunk(Q[l], K[l], sink, start, start + m, out=outs[l], workspace=ws[s], stream=streams[s])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label}: {dt / (6 * L) * 1e6:.1f} us per score_chunk call")


run([torch.cuda.current_stream()], "1 stream (current)")
for trial in range(4):
    run([torch.cuda.Stream(), torch.cuda.Stream()], f"2 default-priority streams (trial {trial})")
run([torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)], "high + normal priority")
run([torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)], "high + high priority")
