#!/bin/bash
# round 3, GPU call 3: decode-graph test, bench (decode via graph), BASELINE C2 through ModelKVzip
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cache.py -m gpu -q -x -k "decode_graph or update_attend" > gpurun_out/r3c3_graph_tests.log 2>&1; echo "graph tests rc=$?"; tail -5 gpurun_out/r3c3_graph_tests.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3c3_bench.json 2> gpurun_out/r3c3_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3c3_bench.json").read().strip().splitlines()[-1])
    print(round(d["value"]), "tok/s", round(d["ms_per_step"],1), "ms; host/pair us", d["config"].get("host_us_per_update_score_pair"), "enqueue ms", round(d["config"]["host_enqueue_ms_per_step"],1))
    print({k:(round(v["avg_ms"]*1e3,1) if v.get("avg_ms") else None) for k,v in d["roofline_stages"].items()})
    print("decode", d["decode"])
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r3c3_bench.err").read()[-2000:])
PY
timeout 900 python tools/e2e_c2.py --json gpurun_out/r3_e2e_c2.json > gpurun_out/r3c3_e2e.log 2>&1; echo "e2e rc=$?"; tail -c 3000 gpurun_out/r3c3_e2e.log
