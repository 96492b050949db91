#!/bin/bash
O=gpurun_out/r2c11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cache.py tests/test_gpu_configs.py -m gpu -x -q -k "attn or cache or c5 or life or retain or attend" > $O/pytest.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python tools/attn_probe.py > $O/attn.txt 2>&1; echo "probe rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest.txt; cat $O/attn.txt
