#!/bin/bash
# round 6 (second session): long fuzz runs on the final tree
O=gpurun_out/r6fz2; mkdir -p $O
timeout 1500 python tools/fuzz_tail.py 300 21 2>&1 | tail -2 > $O/fuzz_tail_long.txt; cat $O/fuzz_tail_long.txt
timeout 1500 python tools/fuzz_prune.py 500 22 2>&1 | tail -1 > $O/fuzz_prune_long.txt; cat $O/fuzz_prune_long.txt
timeout 1500 python tools/fuzz_score.py kvzip_amd/libkvzip_hip.so kvzip_amd/libkvzip_hip.so 300 23 2>&1 | tail -2 > $O/fuzz_score_long.txt; cat $O/fuzz_score_long.txt
