#!/bin/bash
# round 3, GPU call 4: in-kernel timelines (pass A / pass B), host profile, SQ counter passes, C2 through ModelKVzip, new tests
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests -m gpu -q -x -k "propagates_nan or decode_graph or flash2" > $O/r3c4_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r3c4_tests.log
KVZIP_HIP_LIB=$R/tools/ab/lib_trace.so timeout 120 python tools/trace_b.py > $O/r3c4_trace_b.txt 2>&1; echo "trace_b rc=$?"; head -60 $O/r3c4_trace_b.txt | grep -v amdgpu.ids
KVZIP_HIP_LIB=$R/tools/ab/lib_trace.so timeout 120 python tools/trace2.py > $O/r3c4_trace_a.txt 2>&1; echo "trace_a rc=$?"; sed -n 1,40p $O/r3c4_trace_a.txt | grep -v amdgpu.ids
timeout 120 python tools/host_profile.py 3 > $O/r3c4_host_fused.txt 2>&1; head -30 $O/r3c4_host_fused.txt | grep -v amdgpu.ids
timeout 120 python tools/host_profile.py 3 unfused 2>&1 | grep "host " 
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/r3pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/r3pmc2 -o p2 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/r3pmc1 $O/r3pmc2 > $O/r3c4_pmc_sq.json 2>&1; head -c 2500 $O/r3c4_pmc_sq.json
find $O/r3pmc1 $O/r3pmc2 -name "*.csv" -size +2M -delete
timeout 900 python tools/e2e_c2.py --json $O/r3_e2e_c2.json > $O/r3c4_e2e.log 2>&1; echo "e2e rc=$?"; tail -c 2200 $O/r3c4_e2e.log
