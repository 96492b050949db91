#!/bin/bash
# round 6 (second session), final tree (256 candidate-key blocks in the tail): whole GPU suite, smoke, fuzz, profiles, every bench config
O=gpurun_out/r6bp; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "PARITY|E2E|MASK|G15|UNIFORM|passed|failed|Error|error|scores equal" > $O/pytest_gpu.txt; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 900 python tools/fuzz_tail.py 60 2 > $O/fuzz_tail.txt 2>&1; tail -1 $O/fuzz_tail.txt
bash tools/final_profile_r6b.sh prof 2>&1 | tail -30
bash tools/final_profile_r6b.sh bench 2>&1 | tail -13
python bench.py --gpus 1 --steps 20 --warmup 5 --inputs copy > $O/bench_c4_copy_like.json 2> $O/bench_c4_copy_like.err; python -c "
import json; d=json.loads(open('$O/bench_c4_copy_like.json').read().strip().splitlines()[-1]); print('copy-like', round(d['value']), d.get('parity_ok'))"
