#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "score" > gpurun_out/c17_tests.log 2>&1
tail -5 gpurun_out/c17_tests.log
