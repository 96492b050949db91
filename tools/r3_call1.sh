#!/bin/bash
# round 3, GPU call 1: full GPU test-suite on the restructured scoring path, A/B of the one-wave-per-SIMD variants, bench (fused / unfused update)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3c1_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "E2E|passed|failed|Error|error" gpurun_out/r3c1_pytest.log | tail -15
timeout 600 python tools/ab_score.py > gpurun_out/r3c1_ab.log 2>&1; echo "ab rc=$?"
cat gpurun_out/r3c1_ab.log | cut -c1-1500
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3c1_bench_fused.json 2> gpurun_out/r3c1_bench_fused.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --unfused-update > gpurun_out/r3c1_bench_unfused.json 2> gpurun_out/r3c1_bench_unfused.err; echo "bench2 rc=$?"
python - <<'PY'
import json
for n in ("fused","unfused"):
    try:
        d=json.loads(open(f"gpurun_out/r3c1_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"]), "tok/s", round(d["ms_per_step"],1), "ms host", round(d["config"]["host_enqueue_ms_per_step"],1),
              {k:(round(v["avg_ms"]*1e3,1) if v.get("avg_ms") else None) for k,v in d["roofline_stages"].items()}, "decode", round(d["decode"]["ms_per_token"],3))
    except Exception as e:
        print(n, "ERR", e); print(open(f"gpurun_out/r3c1_bench_{n}.err").read()[-1500:])
PY
