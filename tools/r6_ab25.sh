#!/bin/bash
# round 6 (second session): side streams 2 / 3 / 4 with the final tail (pipelined, 256 candidate-key blocks)
O=gpurun_out/r6ah; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2 3; do
  for s in 3 2 4; do echo -n "round $r streams=$s: "; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --score-streams $s 2>/dev/null | python -c "$line"; done
done > $O/ab_streams2.txt 2>&1; cat $O/ab_streams2.txt
