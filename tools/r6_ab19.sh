#!/bin/bash
# round 6 (second session): candidate-key blocks per CU in the pipelined tail: 4 (product: 1 024 blocks) / 3 / 2 / 1, interleaved, four rounds
O=gpurun_out/r6ab; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2 3 4; do
  for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_skb3.so tools/ab/lib_skb2.so tools/ab/lib_skb1.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_skb.txt 2>&1; cat $O/ab_skb.txt
for l in kvzip_amd/libkvzip_hip.so tools/ab/lib_skb2.so tools/ab/lib_skb1.so; do echo -n "bf16 $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --dtype bf16 2>/dev/null | python -c "$line"; done >> $O/ab_skb.txt 2>&1; tail -3 $O/ab_skb.txt
