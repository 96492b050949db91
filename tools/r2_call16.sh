#!/bin/bash
mkdir -p gpurun_out
KVZIP_HIP_LIB=tools/ab/lib_trace.so timeout 300 python tools/trace2.py > gpurun_out/c16_trace.txt 2>&1
grep switch gpurun_out/c16_trace.txt | head -8
timeout 600 python tools/ab_score.py tools/ab/lib_pa2.so kvzip_amd/libkvzip_hip.so > gpurun_out/c16_ab.log 2>&1
