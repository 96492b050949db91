#!/usr/bin/env python
"""Run kvz_score_chunk in a tight loop for a few seconds while polling rocm-smi (power, clocks): is the kernel power-limited?"""
import subprocess
import sys
import threading
import time
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvzip_amd import ops  # noqa: E402

dev = "cuda:0"
H, Hkv, D, sink, N, m = 28, 4, 128, 32, 131072, 2000
q_len = m + 26
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, H, q_len, D, generator=g, device=dev).half()
k = torch.randn(1, Hkv, sink + N + q_len, D, generator=g, device=dev).half()
start = sink + 60000
samples = []
stop = False


def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(s in ln for s in ("Power", "sclk", "mclk", "Temperature (Sensor junction)", "fclk"))]
        samples.append((time.time(), keep))
        time.sleep(0.5)


print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout[-400:])
th = threading.Thread(target=poll)
th.start()
time.sleep(1.5)
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize()
    n += 200
dt = time.time() - t0
time.sleep(1.0)
stop = True
th.join()
print(f"{n} calls in {dt:.2f} s -> {dt / n * 1e6:.1f} us per score_chunk call (all four kernels)")
for ts, keep in samples:
    print(f"t={ts - t0:5.1f}s  " + " | ".join(keep))
