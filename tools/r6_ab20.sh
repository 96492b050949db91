#!/bin/bash
# round 6 (second session): candidate-key blocks of the pipelined tail: 256 (product now) / 512 / 128 / 64 / 32, interleaved; pathological inputs (every pair a candidate) through the test suite
O=gpurun_out/r6ac; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2 3; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_sktb512.so tools/ab/lib_sktb128.so tools/ab/lib_sktb64.so tools/ab/lib_sktb32.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_sktb.txt 2>&1; cat $O/ab_sktb.txt
for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_sktb64.so; do echo -n "copy-like $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 4 --warmup 2 --no-cpu-baseline --decode-tokens 2 --inputs copy 2>/dev/null | python -c "$line"; done >> $O/ab_sktb.txt 2>&1; tail -2 $O/ab_sktb.txt
