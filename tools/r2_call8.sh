#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2c8; mkdir -p $O; cd $GRAFT_REPO_ROOT
rm -f /tmp/ab_oracle_*.pt
timeout 900 python tools/ab_score.py kvzip_amd/libkvzip_hip.so tools/ab/lib_mix16.so tools/ab/lib_noprio.so > $O/ab.txt 2>&1; echo "ab rc=$?" > $O/rc.txt
cat $O/rc.txt
