# rocprofv3 counter passes for the scoring kernels (run on the GPU box: bash tools/pmc_cmd.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_CYCLES --kernel-trace -d $R/gpurun_out/pmc12 -o p12 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc13 -o p13 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc14 -o p14 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
ls $R/gpurun_out/pmc12 $R/gpurun_out/pmc13 $R/gpurun_out/pmc14 | head -3
