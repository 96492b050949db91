#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/score_parity_measured.json
KVZ_RECORD_PARITY=1 timeout 1200 python -m pytest tests -m gpu -q -s > $O/r3c8_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|C2 e2e" $O/r3c8_pytest.log | tail -12
bash tools/final_profile_r3.sh all 2>&1 | tail -30
