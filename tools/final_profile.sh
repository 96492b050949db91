# regenerate profiles/ inputs on the GPU box: bash tools/final_profile.sh   (outputs under gpurun_out/final/)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --score-streams 1 --no-cpu-baseline > $O/bench_1stream.json 2> $O/bench_1stream.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o stats --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --decode-tokens 8 > $O/prof_bench.json 2> $O/prof.err
rocprofv3 --kernel-trace --stats -d $O/prof1 -o stats --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --decode-tokens 8 --score-streams 1 > $O/prof1_bench.json 2> $O/prof1.err
rm -f $O/prof/stats_kernel_trace.csv $O/prof1/stats_kernel_trace.csv
ls $O $O/prof | head -30
