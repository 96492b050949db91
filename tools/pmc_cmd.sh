# rocprofv3 counter passes for the scoring kernels (run on the GPU box: bash tools/pmc_cmd.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc10 -o p10 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc11 -o p11 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
ls $R/gpurun_out/pmc10 $R/gpurun_out/pmc11
