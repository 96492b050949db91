# HBM traffic of the scoring kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes): bash tools/pmc_traffic_score.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcf -o f --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcw -o w --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
ls $R/gpurun_out/pmcf $R/gpurun_out/pmcw | head -4
