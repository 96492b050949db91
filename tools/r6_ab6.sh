#!/bin/bash
# round 6, sixth run: the small launches of a scoring call as few fat 1024-thread blocks (product build, KVZ_PACK=4) vs 256-thread blocks (lib_pack1),
# and the dense append alone (lib_pack4_a256: packed scoring kernels, 256-thread append)
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_prune_path.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -3 $O/pytest_prune.txt
PRUNE_VARIANTS=1,3,4 timeout 900 python tools/proto/prune_check.py 2>&1 | grep -E "^shape|prune=|rror" > $O/prune_check_f16.txt; cut -c1-200 $O/prune_check_f16.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d["roofline_stages"]; print(round(d["value"]), round(d["ms_per_step"],1), "A", round(st["score_rowstat"]["avg_ms"]*1e3,1), "bounds", round(st["score_bounds"]["avg_ms"]*1e3,1), "B", round(st["score_colmax"]["avg_ms"]*1e3,1))'
for r in 1 2 3; do
  for l in tools/ab/lib_pack1.so tools/ab/lib_pack4_a256.so kvzip_amd/libkvzip_hip.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt
for s in 2 4; do echo -n "product, $s streams: "; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --score-streams $s 2>/dev/null | python -c "$line"; done > $O/ab_streams.txt 2>&1; cat $O/ab_streams.txt
for l in tools/ab/lib_pack1.so kvzip_amd/libkvzip_hip.so; do echo -n "bf16 $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done > $O/ab_bench_bf16.txt 2>&1; cat $O/ab_bench_bf16.txt
