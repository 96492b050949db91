#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/ab_score.py > gpurun_out/c15_ab.log 2>&1
cat gpurun_out/c15_ab.log
