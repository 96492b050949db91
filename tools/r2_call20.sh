#!/bin/bash
mkdir -p gpurun_out
for n in 2 3 4 2 3; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode-tokens 2 --score-streams $n > gpurun_out/c20_s$n.json 2> gpurun_out/c20_s$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c20_s$n.json").read().strip().splitlines()[-1])
    st = d["roofline_stages"]
    print("streams $n", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms  rowstat", round(st["score_rowstat"]["avg_ms"] * 1e3, 1), "colmax", round(st["score_colmax"]["avg_ms"] * 1e3, 1), "host", round(d["config"]["host_enqueue_ms_per_step"], 1))
except Exception as e:
    print("$n failed", e)
PY
done
