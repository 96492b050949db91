#!/bin/bash
# round 6: bounds kernel diet (bounds converted once, DPP quad prefix, bit-loop stores) vs the committed build
O=gpurun_out/r6j; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -2 $O/pytest_prune.txt
PRUNE_VARIANTS=1,3,4 timeout 900 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|prune=3|rror" | cut -c1-230 > $O/prune_check_f16.txt; cat $O/prune_check_f16.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d["roofline_stages"]; print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2), "bounds", round(st["score_bounds"]["avg_ms"]*1e3,1))'
for r in 1 2 3; do
  for l in tools/ab/lib_head.so kvzip_amd/libkvzip_hip.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt
