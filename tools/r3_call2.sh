#!/bin/bash
# round 3, GPU call 2: GPU suite with parity recording (measured values -> tests/golden/score_parity_measured.json), new flash kernel tests + probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "flash2" -x > gpurun_out/r3c2_flash2_tests.log 2>&1; echo "flash2 tests rc=$?"
grep -E "flash2 \(|passed|failed|Error" gpurun_out/r3c2_flash2_tests.log | tail -20
timeout 300 python tools/flash2_probe.py > gpurun_out/r3c2_flash2_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3c2_flash2_probe.txt | grep -v amdgpu.ids
rm -f gpurun_out/score_parity_measured.json
KVZ_RECORD_PARITY=1 timeout 900 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_e2e_c2.py > gpurun_out/r3c2_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^E2E|E2E D|passed|failed|^FAILED|^ERROR" gpurun_out/r3c2_pytest.log | tail -20
