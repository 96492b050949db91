#!/bin/bash
# Round-2 final measurements on the GPU box (bash tools/final_profile_r2.sh [part]); outputs under gpurun_out/r2final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2final; mkdir -p $O; cd $R
part=${1:-all}
if [ $part = bench -o $part = all ]; then
  python bench.py --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?" > $O/rc.txt
  python bench.py --steps 5 --warmup 2 --ctx 32768 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.txt
  python bench.py --steps 3 --warmup 1 --model llama3.1-8b > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --model qwen2.5-14b --level head --dtype bf16 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; echo "c4bf16 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --score-streams 1 --no-cpu-baseline > $O/bench_c4_1stream.json 2> $O/bench_c4_1stream.err; echo "c4 1stream rc=$?" >> $O/rc.txt
  cat $O/rc.txt
fi
if [ $part = prof -o $part = all ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $O/prof1 -o stats --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --decode-tokens 8 --score-streams 1 > $O/prof1_bench.json 2> $O/prof1.err
  rocprofv3 --kernel-trace --stats -d $O/prof3 -o stats --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --decode-tokens 8 > $O/prof3_bench.json 2> $O/prof3.err
  rm -f $O/prof1/*kernel_trace.csv $O/prof3/*kernel_trace.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf -o f --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcw -o w --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc2 -o p2 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcfa -o f --output-format csv -- python $R/tools/prof_score.py attn 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcwa -o w --output-format csv -- python $R/tools/prof_score.py attn 3 > /dev/null 2>&1
  ls $O
fi
