cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in fixedtile noepi_nomfma noepi_fixedtile; do KVZIP_HIP_LIB=$R/tools/ab/lib_$v.so python $R/tools/prof_score_kernels.py 20 2>&1 | tail -1; done
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/pmc7 -o p7 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc8 -o p8 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace -d $R/gpurun_out/pmc9 -o p9 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
ls $R/gpurun_out/pmc7 $R/gpurun_out/pmc8 $R/gpurun_out/pmc9
