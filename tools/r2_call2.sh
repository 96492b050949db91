#!/bin/bash
# GPU call 2: A/B of the software-pipelined scoring kernels (parity vs oracle, determinism, kernel times), then the GPU suite
O=gpurun_out/r2c2; mkdir -p $O
timeout 1100 python tools/ab_score.py kvzip_amd/libkvzip_hip.so tools/ab/lib_r1.so tools/ab/lib_s0.so tools/ab/lib_w4.so tools/ab/lib_a2b1.so tools/ab/lib_a1b2.so > $O/ab.txt 2>&1
echo "ab rc=$?" > $O/rc.txt
timeout 600 python -m pytest tests -m gpu -x -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/ab.txt | cut -c1-600; tail -5 $O/pytest.txt
