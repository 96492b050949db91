#!/bin/bash
# round 6 (second session), final tree: fuzz of the pipelined tail, kernel stats / timeline / counters, every bench config
O=gpurun_out/r6bp; mkdir -p $O
timeout 900 python tools/fuzz_tail.py 60 1 > $O/fuzz_tail.txt 2>&1; tail -3 $O/fuzz_tail.txt
timeout 900 python tools/fuzz_prune.py 150 7 2>&1 | tail -1 > $O/fuzz_prune.txt; cat $O/fuzz_prune.txt
bash tools/final_profile_r6b.sh prof 2>&1 | tail -40
bash tools/final_profile_r6b.sh bench 2>&1 | tail -16
