#!/bin/bash
# (time-attribution builds: apply tools/attic/abl_keys_r6.patch to kvzip_amd/csrc first, then tools/ab_build.sh skablN -DKVZ_SK_ABL=N)
# round 6 (second session): candidate-key pass with the streaming cache policy (nt) on its query-row DMA (1), its key-row gather (2), both (3)
O=gpurun_out/r6w; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
python -m pytest tests/test_gpu_tail_pipeline.py -x -q -m gpu 2>&1 | tail -1
for r in 1 2 3; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_sknt1.so tools/ab/lib_sknt2.so tools/ab/lib_sknt3.so tools/ab/lib_skabl5.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_sknt.txt 2>&1; cat $O/ab_sknt.txt
