#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/host_profile.py 3 2>&1 | grep "host "
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q -k "async or stream or cache or life or model or multi" 2>&1 | tail -3
