#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/ab_score.py tools/ab/lib_bfold.so kvzip_amd/libkvzip_hip.so > gpurun_out/c22_ab.log 2>&1
