#!/bin/bash
# (time-attribution builds: apply tools/attic/abl_keys_r6.patch to kvzip_amd/csrc first, then tools/ab_build.sh skablN -DKVZ_SK_ABL=N)
# round 6 (second session): the candidate-key items part by part, REPEATED (the builds of r6_ab13 / 16 / 17 had lost their candidate-key blocks altogether: a macro of the
# attribution itself; those numbers are "no candidate-key blocks", not what their names say).  5 = query rows not fetched, 6 = key rows not fetched, 7 = no atomics (and, dead, no
# compute), 9 / 10 = half of the query rows / of each key row, 11 = plain stores, 13 = plain load first + only atomics that can change the value (correct), 14 = atomics that change nothing
O=gpurun_out/r6aa; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
KVZIP_HIP_LIB=$PWD/tools/ab/lib_skabl13.so python -m pytest tests/test_gpu_tail_pipeline.py tests/test_gpu_prune_path.py -x -q -m gpu 2>&1 | tail -1
for r in 1 2; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_skabl5.so tools/ab/lib_skabl6.so tools/ab/lib_skabl7.so tools/ab/lib_skabl9.so tools/ab/lib_skabl10.so tools/ab/lib_skabl11.so tools/ab/lib_skabl13.so tools/ab/lib_skabl14.so tools/ab/lib_skabl3.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_items.txt 2>&1; cat $O/ab_items.txt
