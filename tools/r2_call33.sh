#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "score" 2>&1 | tail -2
timeout 600 python tools/ab_score.py tools/ab/lib_old.so tools/ab/lib_new.so > gpurun_out/c33_ab.log 2>&1
bash tools/r2_call21.sh old new
