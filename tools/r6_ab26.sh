#!/bin/bash
# (the KVZ_TAIL_SERIAL variant was removed from the source after this run)
# round 6 (second session): tail launch with 256 blocks that each run their share of ALL three phases one after the other (tser) against one phase per block (product: 732 blocks)
O=gpurun_out/r6ai; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
KVZIP_HIP_LIB=$PWD/tools/ab/lib_tser.so python -m pytest tests/test_gpu_tail_pipeline.py -x -q -m gpu 2>&1 | tail -1
for r in 1 2 3; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_tser.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_tser.txt 2>&1; cat $O/ab_tser.txt
