#!/bin/bash
# bench (overlapped, 2 streams): key-slice length of pass A / hidden-tile skip
mkdir -p gpurun_out
timeout 300 python tools/ab_score.py tools/ab/lib_ks8skip.so > gpurun_out/c19_ab.log 2>&1
grep -o '"headline_f16": {[^}]*}\|"spiky_f16": {[^}]*}\|"negative_f16": {[^}]*}\|"d64_f16": {[^}]*}\|"d128_bf16": {[^}]*}' gpurun_out/c19_ab.log | head -5
for v in ks8 ks8skip ks8 ks8skip; do
  KVZIP_HIP_LIB=tools/ab/lib_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode-tokens 2 > gpurun_out/c19_$v.json 2> gpurun_out/c19_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c19_$v.json").read().strip().splitlines()[-1])
    st = d["roofline_stages"]
    print("$v", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms  rowstat", round(st["score_rowstat"]["avg_ms"] * 1e3, 1), "colmax", round(st["score_colmax"]["avg_ms"] * 1e3, 1))
except Exception as e:
    print("$v failed", e)
PY
done
