#!/bin/bash
O=gpurun_out/r2c5; mkdir -p $O
timeout 900 python tools/ab_score.py kvzip_amd/libkvzip_hip.so tools/ab/lib_r1.so tools/ab/lib_ks8.so tools/ab/lib_ks2.so > $O/ab.txt 2>&1; echo "ab rc=$?" > $O/rc.txt
KVZIP_HIP_LIB=tools/ab/lib_trace.so timeout 200 python tools/trace2.py > $O/trace.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
cat $O/rc.txt
