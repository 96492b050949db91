#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cache.py tests/test_gpu_model.py -m gpu -q -x -s -k "fused_statistics or model or prefill or evict_and_retain" > $O/r3c7_tests.log 2>&1; echo "tests rc=$?"; grep -E "fused vs|PARITY fused|passed|failed|Error" $O/r3c7_tests.log | tail -8
timeout 300 python tools/flash2_probe.py --no-sdpa 2>&1 | grep -v amdgpu.ids | tail -6
