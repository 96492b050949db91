#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/ab_score.py tools/ab/lib_snake.so tools/ab/lib_plan.so > gpurun_out/c30_ab.log 2>&1
bash tools/r2_call21.sh snake plan
