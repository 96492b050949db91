#!/bin/bash
# round 6, second run: bf16 on the pruned call (checks + times), whole GPU suite, bench lines of both dtypes vs the round-5 library
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -3 $O/pytest_prune.txt
PRUNE_DTYPE=bf16 PRUNE_VARIANTS=0,1,3 timeout 600 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|prune=|rror" > $O/prune_check_bf16.txt; cat $O/prune_check_bf16.txt
PRUNE_VARIANTS=0,1,3 timeout 600 python tools/proto/prune_check.py 2>&1 | grep -E "prune=" | head -9 > $O/prune_times_f16.txt; cat $O/prune_times_f16.txt
BENCH_FLAGS="--dtype bf16" timeout 900 bash tools/ab_bench.sh 2 $PWD/tools/ab/lib_t2_r5.so $PWD/kvzip_amd/libkvzip_hip.so > $O/ab_bench_bf16.txt 2>&1; cat $O/ab_bench_bf16.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
