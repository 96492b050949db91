#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2c7; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_score.py kvzip_amd/libkvzip_hip.so tools/ab/lib_prio.so tools/ab/lib_w4.so > $O/ab.txt 2>&1; echo "ab rc=$?" > $O/rc.txt
KVZIP_HIP_LIB=tools/ab/lib_w4t.so timeout 200 python tools/trace2.py > $O/trace_w4.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
KVZIP_HIP_LIB=tools/ab/lib_prio2.so timeout 200 python tools/trace2.py > $O/trace_prio.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
cat $O/rc.txt
