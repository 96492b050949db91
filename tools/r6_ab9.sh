#!/bin/bash
# round 6 (second session): rounding chain through v_fma_mixlo_f16 / v_fma_mixhi_f16 (product) vs v_fma_mix_f32 + v_cvt_pk_f16_f32 (lib_nomix);
# side-stream counts 3 / 4 / 5 with 8 hardware queues (HIP multiplexes its streams on GPU_MAX_HW_QUEUES = 4 by default: 3 side streams + the caller's fill them)
O=gpurun_out/r6r; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "round_chain" > $O/pytest_chain.txt 2>&1; tail -3 $O/pytest_chain.txt
python -m pytest tests/test_gpu_prune_path.py tests/test_gpu_far_context.py -x -q -m gpu > $O/pytest_prune.txt 2>&1; tail -2 $O/pytest_prune.txt
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/1848,2), "parity", d.get("parity_ok"))'
for r in 1 2 3; do
  for l in tools/ab/lib_nomix.so kvzip_amd/libkvzip_hip.so; do echo -n "round $r $(basename $l): "; KVZIP_HIP_LIB=$PWD/$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"; done
done > $O/ab_mixlo.txt 2>&1; cat $O/ab_mixlo.txt
for r in 1 2; do
  for cfg in "4:3" "8:3" "8:4" "8:5" "8:6" "16:4"; do
    q=${cfg%%:*}; s=${cfg##*:}
    echo -n "round $r hwq=$q streams=$s: "; GPU_MAX_HW_QUEUES=$q python bench.py --steps 6 --warmup 2 --no-cpu-baseline --decode-tokens 2 --score-streams $s 2>/dev/null | python -c "$line"
  done
done > $O/ab_streams.txt 2>&1; cat $O/ab_streams.txt
