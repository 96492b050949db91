#!/bin/bash
# round 6 (second session): pipelined tail of the pruned call (score_prune = 6: merge of call i + bounds of call i-1 + candidate keys of call i-2 of a stream
# in ONE launch) against the chained call (3), same box, interleaved; 2 / 3 side streams
O=gpurun_out/r6s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tail_pipeline.py -x -q -m gpu > $O/pytest_tail.txt 2>&1; tail -15 $O/pytest_tail.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2), "parity", d.get("parity_sample",{}).get("parity_ok"), "host ms", round(d.get("host_enqueue_ms_per_step",0),1))'
for r in 1 2 3; do
  for cfg in "3:3" "6:3" "6:2"; do
    k=${cfg%%:*}; s=${cfg##*:}
    echo -n "round $r score_prune=$k streams=$s: "; timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --tune score_prune=$k --score-streams $s 2>$O/err_$k_$s.txt | python -c "$line"
  done
done > $O/ab_tail.txt 2>&1; cat $O/ab_tail.txt; tail -3 $O/err_*.txt
