#!/bin/bash
# round 3, GPU call 6: forward-fused statistics (f2): tests, C2 through ModelKVzip both ways; pass B prologue timing
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cache.py tests/test_gpu_model.py -m gpu -q -x -s -k "fused_statistics or model or prefill or evict_and_retain" > $O/r3c6_tests.log 2>&1; echo "tests rc=$?"; grep -E "fused vs|PARITY fused|passed|failed|Error" $O/r3c6_tests.log | tail -8
timeout 300 python tools/ab_score.py $R/kvzip_amd/libkvzip_hip.so > $O/r3c6_ab.log 2>&1; grep -o '"time_f16": {[^}]*}' $O/r3c6_ab.log
timeout 900 python tools/e2e_c2.py --json $O/r3_e2e_c2.json > $O/r3c6_e2e.log 2>&1; echo "e2e rc=$?"; tail -c 1800 $O/r3c6_e2e.log
timeout 900 python tools/e2e_c2.py --two-pass --no-oracle --json $O/r3_e2e_c2_two_pass.json > $O/r3c6_e2e2.log 2>&1; echo "e2e two-pass rc=$?"; tail -c 900 $O/r3c6_e2e2.log
