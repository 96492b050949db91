#!/bin/bash
# multi-row kernel: tests, crossover probe, dense forward vs SDPA
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "many_query_rows or flash_fwd or varlen_attn" > gpurun_out/c13_tests.log 2>&1
tail -5 gpurun_out/c13_tests.log
timeout 600 python tools/flash_probe.py > gpurun_out/c13_probe.log 2>&1
cat gpurun_out/c13_probe.log
