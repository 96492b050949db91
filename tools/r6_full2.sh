#!/bin/bash
# round 6 (second session): whole GPU suite with the default knob and with the pipelined tail preset (KVZIP_SCORE_PRUNE=6), smoke, host cost of a call pair
# for both, the driver's bench command
O=gpurun_out/r6full2; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "PARITY|E2E|MASK|G15|UNIFORM|passed|failed|Error|error|scores equal" > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
KVZIP_SCORE_PRUNE=6 timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/pytest_gpu_knob6.txt; cat $O/pytest_gpu_knob6.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
for k in 3 6; do echo "score_prune=$k: $(KVZIP_SCORE_PRUNE=$k python tools/host_profile.py 3 2>/dev/null | head -1)"; done > $O/host_profile.txt; cat $O/host_profile.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6full2/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'parity_ok', d.get('parity_ok'))
print('host', d['config']['host_enqueue_ms_per_step'], d['config']['host_us_per_update_score_pair'])
PY
KVZIP_SCORE_PRUNE=6 python bench.py > $O/bench_knob6.json 2> $O/bench_knob6.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6full2/bench_knob6.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'parity_ok', d.get('parity_ok'))
print('host', d['config']['host_enqueue_ms_per_step'], d['config']['host_us_per_update_score_pair'])
PY
