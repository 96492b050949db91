#!/bin/bash
# is the first bench.py process on a fresh box slower?  (side-stream choice logged)
mkdir -p gpurun_out
run() {
  KVZ_STREAM_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup $2 --no-cpu-baseline --decode-tokens 2 > gpurun_out/frp.json 2> gpurun_out/frp.err
  grep "side streams" gpurun_out/frp.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/frp.json").read().strip().splitlines()[-1])
print("$1 warmup $2:", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms")
PY
}
run first ${1:-1}; run second 1; run third 1
