#!/bin/bash
# GPU call 1 of round 2: test suite + bench lines of every BASELINE config (round-1 kernels, new harness)
O=gpurun_out/r2c1; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
python bench.py --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?" >> $O/rc.txt
python bench.py --steps 5 --warmup 2 --ctx 32768 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.txt
python bench.py --steps 3 --warmup 1 --model llama3.1-8b > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.txt
python bench.py --steps 5 --warmup 2 --model qwen2.5-14b --level head --dtype bf16 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.txt
python bench.py --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; echo "c4bf16 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest.txt
