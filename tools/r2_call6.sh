#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2c6; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_score.py kvzip_amd/libkvzip_hip.so tools/ab/lib_r1.so tools/ab/lib_ks8.so > $O/ab.txt 2>&1; echo "ab rc=$?" > $O/rc.txt
KVZIP_HIP_LIB=tools/ab/lib_trace.so timeout 200 python tools/trace2.py > $O/trace.txt 2>&1; echo "trace rc=$?" >> $O/rc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > $O/pmc1.log 2>&1
cat $O/rc.txt
