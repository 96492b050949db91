#!/bin/bash
# round 6 (second session): repeat-prompt-like inputs (bench.py --inputs copy): the pruned call (default) against the two-pass call, parity sample on the same kind of inputs
O=gpurun_out/r6t; mkdir -p $O
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2), "parity_ok", d.get("parity_ok"), {k:v for k,v in (d.get("parity_sample") or {}).items() if k in ("f16","bf16","violations")})'
python bench.py --inputs copy --steps 4 --warmup 1 > $O/bench_copy_default.json 2> $O/bench_copy_default.err; echo "rc=$?"; python -c "$line" < $O/bench_copy_default.json
python bench.py --inputs copy --steps 4 --warmup 1 --dtype bf16 > $O/bench_copy_bf16.json 2> $O/bench_copy_bf16.err; echo "rc=$?"; python -c "$line" < $O/bench_copy_bf16.json
for r in 1 2; do
  for k in 6 0 5; do echo -n "round $r copy-like score_prune=$k: "; python bench.py --inputs copy --steps 4 --warmup 1 --no-cpu-baseline --decode-tokens 2 --tune score_prune=$k 2>/dev/null | python -c "$line"; done
done > $O/ab_copy.txt 2>&1; cat $O/ab_copy.txt
tail -3 $O/*.err
