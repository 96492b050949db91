#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cache.py tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5
timeout 200 python tools/host_profile.py 3 2>&1 | grep "host "
for i in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --decode-tokens 2 > gpurun_out/c27.json 2> gpurun_out/c27.err
python - <<PY
import json
d = json.loads(open("gpurun_out/c27.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms host", round(d["config"]["host_enqueue_ms_per_step"], 1))
PY
done
