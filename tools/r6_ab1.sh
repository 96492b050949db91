#!/bin/bash
# round 6, first A/B: wave-uniform reference (product build) and the per-block mask variants vs the round-5 library
O=gpurun_out/r6a; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -3 $O/pytest_prune.txt
for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_t2_bm1.so tools/ab/lib_t2_bm2.so; do
  echo "== prune_check $l"; KVZIP_HIP_LIB=$PWD/$l PRUNE_VARIANTS=0,1,3 PRUNE_NOTIME=1 timeout 600 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|Error|error" 
done > $O/prune_check.txt 2>&1
cat $O/prune_check.txt
timeout 900 python tools/proto/t2_time.py 3 > $O/t2_time.txt 2>&1; cat $O/t2_time.txt
BENCH_FLAGS="" timeout 1200 bash tools/ab_bench.sh 2 $PWD/tools/ab/lib_t2_r5.so $PWD/kvzip_amd/libkvzip_hip.so $PWD/tools/ab/lib_t2_bm1.so $PWD/tools/ab/lib_t2_bm2.so > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt
