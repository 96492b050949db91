#!/bin/bash
# round 6: candidate-key pass without the list-length load (product), its grid (lib_sk2 / lib_sk1: 512 / 256 blocks instead of 1024), dense append with 64 blocks (lib_appbx16)
O=gpurun_out/r6h; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -2 $O/pytest_prune.txt
PRUNE_VARIANTS=1,3,4 PRUNE_NOTIME=1 timeout 900 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|rror" | cut -c1-210 > $O/prune_check_f16.txt; cat $O/prune_check_f16.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2 3; do
  for l in tools/ab/lib_head.so kvzip_amd/libkvzip_hip.so tools/ab/lib_sk2.so tools/ab/lib_sk1.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt
