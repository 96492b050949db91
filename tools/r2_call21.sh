#!/bin/bash
# bench (overlapped) for library variants given as arguments (names of tools/ab/lib_<name>.so), two rounds
mkdir -p gpurun_out
for rnd in 0 1; do
for v in "$@"; do
  KVZIP_HIP_LIB=tools/ab/lib_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode-tokens 2 > gpurun_out/c21_$v.json 2> gpurun_out/c21_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c21_$v.json").read().strip().splitlines()[-1])
    st = d["roofline_stages"]
    print("round $rnd $v", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms  rowstat", round(st["score_rowstat"]["avg_ms"] * 1e3, 1), "colmax", round(st["score_colmax"]["avg_ms"] * 1e3, 1))
except Exception as e:
    print("$v failed", e)
PY
done
done
