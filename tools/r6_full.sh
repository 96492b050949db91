#!/bin/bash
# round 6: whole GPU suite (with the parity lines), smoke, the driver's bench command
O=gpurun_out/r6full; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "PARITY|E2E|MASK|G15|UNIFORM|passed|failed|Error|error|scores equal" > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6full/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'parity_ok', d.get('parity_ok'))
st=d['roofline_stages']
print({k:(round(v['avg_ms']*1e3,1) if v and v.get('avg_ms') else None) for k,v in st.items()})
print('compact', {k:st['compact_gather'].get(k) for k in ('frac','frac_of_box_copy','frac_of_box_copy_kernel','hbm_copy_GBps_this_box','hbm_copy_kernel_GBps_this_box')})
print('select_compact', st.get('select_compact'))
print('decode', d['decode']['ms_per_token'], d['decode']['ms_per_token_hip_graph'], st['decode_varlen_attn']['frac'], st['decode_varlen_attn'].get('frac_loop'))
print('host', d['config']['host_enqueue_ms_per_step'], d['config']['host_us_per_update_score_pair'])
print('roofline', {k:d['roofline'].get(k) for k in ('achieved','frac','avg_ms','achieved_step_tflops','traffic')})
print('parity', d.get('parity_sample'))
PY
