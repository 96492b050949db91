#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 2 > gpurun_out/c28.json 2> gpurun_out/c28.err
python - <<PY
import json
d = json.loads(open("gpurun_out/c28.json").read().strip().splitlines()[-1])
print("bench $i", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms host", round(d["config"]["host_enqueue_ms_per_step"], 1), d["config"].get("score_streams_distinct"))
PY
done
