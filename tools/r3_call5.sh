#!/bin/bash
# round 3, GPU call 5: pass B with the two-round-trip prologue and the LDS-free epilogue: parity, timing, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e_parity.py tests/test_gpu_cache.py -m gpu -q -x -k "score or e2e or async or deferred" > $O/r3c5_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r3c5_tests.log
timeout 300 python tools/ab_score.py $R/kvzip_amd/libkvzip_hip.so > $O/r3c5_ab.log 2>&1; grep -o '"time_f16": {[^}]*}' $O/r3c5_ab.log
KVZIP_HIP_LIB=$R/tools/ab/lib_trace.so timeout 120 python tools/trace_b.py > $O/r3c5_trace_b.txt 2>&1; grep -v amdgpu.ids $O/r3c5_trace_b.txt | head -24
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r3c5_bench.json 2> $O/r3c5_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3c5_bench.json").read().strip().splitlines()[-1])
    print(round(d["value"]), "tok/s", round(d["ms_per_step"],1), "ms; host/pair us", d["config"].get("host_us_per_update_score_pair"))
    print({k:(round(v["avg_ms"]*1e3,1) if v.get("avg_ms") else None) for k,v in d["roofline_stages"].items()})
    print("decode", d["decode"]["ms_per_token"], d["decode"]["ms_per_token_hip_graph"])
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r3c5_bench.err").read()[-2000:])
PY
