#!/bin/bash
# GPU call 3: pipeline micro-probe (what a SIMD makes of MFMA + epilogue streams) + PMC passes of the v2 scoring kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r2c3; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 tools/probe_pipe > $O/probe_pipe.txt 2>&1; echo "probe rc=$?" > $O/rc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc2 -o p2 --output-format csv -- python $R/tools/prof_score.py score 3 > $O/pmc2.log 2>&1
echo "pmc rc=$?" >> $O/rc.txt
find $O -name "*counter_collection.csv" | head; cat $O/rc.txt; cat $O/probe_pipe.txt
