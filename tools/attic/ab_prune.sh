#!/bin/bash
# same-box A/B of the headline bench line: knob score_prune off / on (prototype build), interleaved: tools/proto/ab_prune.sh <rounds>
rounds=${1:-2}
for r in $(seq 1 $rounds); do
  for pr in 0 3; do
    echo -n "round $r score_prune=$pr: "
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --tune score_prune=$pr $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline_stages']; print(round(d['value']), round(d['ms_per_step'],1), 'A', round(st['score_rowstat']['avg_ms']*1e3,1), 'B', round(st['score_colmax']['avg_ms']*1e3,1), 'parity', json.dumps(d.get('parity_sample'))[:160])"
  done
done
