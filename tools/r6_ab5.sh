#!/bin/bash
# round 6, fifth run: candidates at key granularity (score_prune=3) vs candidate pairs (score_prune=5)
O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -3 $O/pytest_prune.txt
PRUNE_VARIANTS=1,3,4,5 timeout 900 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|prune=|rror" > $O/prune_check_f16.txt; cat $O/prune_check_f16.txt
PRUNE_DTYPE=bf16 PRUNE_VARIANTS=1,3,5 timeout 600 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|prune=|rror" > $O/prune_check_bf16.txt; tail -12 $O/prune_check_bf16.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d["roofline_stages"]; print(round(d["value"]), round(d["ms_per_step"],1), "A", round(st["score_rowstat"]["avg_ms"]*1e3,1), "bounds", round(st["score_bounds"]["avg_ms"]*1e3,1), "B", round(st["score_colmax"]["avg_ms"]*1e3,1))'
for r in 1 2; do
  for pr in 5 3; do echo -n "round $r score_prune=$pr: "; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --tune score_prune=$pr 2>/dev/null | python -c "$line"; done
done > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt
for pr in 5 3; do echo -n "bf16 score_prune=$pr: "; python bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --tune score_prune=$pr 2>/dev/null | python -c "$line"; done > $O/ab_bench_bf16.txt 2>&1; cat $O/ab_bench_bf16.txt
for pr in 3; do echo -n "2 streams score_prune=$pr: "; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --score-streams 2 --tune score_prune=$pr 2>/dev/null | python -c "$line"; done > $O/ab_bench_2s.txt 2>&1; cat $O/ab_bench_2s.txt
for r in 1 2; do echo -n "sk2 (2 blocks per CU) round $r: "; KVZIP_HIP_LIB=$PWD/tools/ab/lib_sk2.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
