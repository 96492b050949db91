#!/bin/bash
# round 6: what each small launch of a scoring call costs IN THE LOOP (three side streams): launches left out one by one (score_prune = 16 + mask;
# results unusable, the step time is the measurement), and the dense append with fewer blocks
O=gpurun_out/r6g; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2))'
for r in 1 2; do
  for pr in 3 20 22 23; do echo -n "round $r score_prune=$pr: "; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --tune score_prune=$pr 2>$O/err_$pr.txt | python -c "$line" || tail -5 $O/err_$pr.txt; done
done > $O/ablate.txt 2>&1; cat $O/ablate.txt
