#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
  for prio in 1 0; do
    for cfg in "--model llama3.1-8b" ""; do
      KVZ_STREAM_PRIO=$prio timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode-tokens 2 $cfg > gpurun_out/c26.json 2> gpurun_out/c26.err
      python - <<PY
import json
d = json.loads(open("gpurun_out/c26.json").read().strip().splitlines()[-1])
print("run $i prio $prio [$cfg]", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms host", round(d["config"]["host_enqueue_ms_per_step"], 1), "streams", d["config"].get("score_streams_distinct"))
PY
    done
  done
done
