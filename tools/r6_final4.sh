#!/bin/bash
# round 6 (second session): last check of the committed tree: GPU suite, smoke, the driver's bench command
O=gpurun_out/r6fin; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_driver_style.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), 'parity_ok', d.get('parity_ok'), 'decode', round(d['decode']['ms_per_token'],3), 'host', round(d['config']['host_us_per_update_score_pair'],1))"
