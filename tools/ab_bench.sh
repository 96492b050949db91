#!/bin/bash
# same-box A/B of the headline bench line for library builds: tools/ab_bench.sh <rounds> <lib.so> [<lib.so> ...]  (extra bench flags in $BENCH_FLAGS)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for l in "$@"; do
    echo -n "round $r $(basename $l): "
    KVZIP_HIP_LIB=$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline_stages']; print(round(d['value']), round(d['ms_per_step'],1), 'A', round(st['score_rowstat']['avg_ms']*1e3,1), 'B', round(st['score_colmax']['avg_ms']*1e3,1))"
  done
done
