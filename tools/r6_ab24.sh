#!/bin/bash
# (the static form was dropped after this run: the source no longer has KVZ_SK_STATIC)
# round 6 (second session): candidate-key pass without the work queue (product: static shares of the (head, group) pairs, rows staged once per group, two round trips before the
# matrix work instead of four; bounds blocks without the queue position) against the queue form (lib_skq.so)
O=gpurun_out/r6ag; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py tests/test_gpu_tail_pipeline.py tests/test_gpu_far_context.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/fuzz_prune.py 120 11 2>&1 | tail -1
timeout 600 python tools/fuzz_tail.py 40 5 2>&1 | tail -1
PRUNE_VARIANTS=1,3,4 PRUNE_NOTIME=1 timeout 900 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|rror" | cut -c1-200 > $O/prune_check.txt; cat $O/prune_check.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2 3; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_skq.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_static.txt 2>&1
for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_skq.so; do echo -n "bf16 $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --dtype bf16 2>/dev/null | python -c "$line"; echo -n "copy-like $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 4 --warmup 2 --no-cpu-baseline --decode-tokens 2 --inputs copy 2>/dev/null | python -c "$line"; done >> $O/ab_static.txt 2>&1; cat $O/ab_static.txt
