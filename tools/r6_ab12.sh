#!/bin/bash
# (time-attribution builds: apply tools/attic/abl_keys_r6.patch to kvzip_amd/csrc first, then tools/ab_build.sh skablN -DKVZ_SK_ABL=N)
# round 6 (second session): where the time of the candidate-key phase goes IN THE LOOP (pipelined tail; time-attribution builds, results unusable):
# product / skabl2: queue entry + candidate list loaded, nothing computed / skabl1: blocks load the counters and leave / skabl3: no candidate-key blocks / skabl4: no bounds blocks either
O=gpurun_out/r6u; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2 3; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_skabl2.so tools/ab/lib_skabl1.so tools/ab/lib_skabl3.so tools/ab/lib_skabl4.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_skabl.txt 2>&1; cat $O/ab_skabl.txt
