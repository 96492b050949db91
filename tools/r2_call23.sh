#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -s -k "hamming or headline_shape or c1 or C1 or config" > gpurun_out/c23_parity.txt 2>&1
grep -i "hamming\|bit-identical\|passed\|failed" gpurun_out/c23_parity.txt
