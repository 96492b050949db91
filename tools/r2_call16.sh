#!/bin/bash
mkdir -p gpurun_out
KVZIP_HIP_LIB=tools/ab/lib_trace.so timeout 300 python tools/trace2.py > gpurun_out/c16_trace.txt 2>&1
grep -c switch gpurun_out/c16_trace.txt
