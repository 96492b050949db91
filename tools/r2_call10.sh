#!/bin/bash
O=gpurun_out/r2c10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
python bench.py --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?" >> $O/rc.txt
python bench.py --steps 5 --warmup 2 --score-streams 1 --no-cpu-baseline > $O/bench_c4_1s.json 2> $O/bench_c4_1s.err; echo "c4 1stream rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest.txt
