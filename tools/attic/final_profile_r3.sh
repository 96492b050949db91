#!/bin/bash
# Round-3 measurements on the GPU box (bash tools/final_profile_r3.sh [bench|prof|pmc|e2e|all]); outputs under gpurun_out/r3final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O; cd $R
export TMPDIR=/tmp
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
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/prof1 -o stats --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --decode-tokens 8 --score-streams 1 > $O/prof1_bench.json 2> $O/prof1.err
  rocprofv3 --kernel-trace --stats -d $O/prof3 -o stats --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --decode-tokens 8 > $O/prof3_bench.json 2> $O/prof3.err
  rm -f $O/prof1/*kernel_trace.csv $O/prof3/*kernel_trace.csv $O/prof1/*agent_info.csv $O/prof3/*agent_info.csv
  ls $O/prof1 $O/prof3
  cd $R
fi
if [ $part = pmc -o $part = all ]; then
  cd /tmp
  for what in score attn; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf_$what -o f --output-format csv -- python $R/tools/prof_score.py $what 3 > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcw_$what -o w --output-format csv -- python $R/tools/prof_score.py $what 3 > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc2 -o p2 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf_flash -o f --output-format csv -- python $R/tools/prof_score.py flash 2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace -d $O/pmc_flash -o p --output-format csv -- python $R/tools/prof_score.py flash 2 > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmcf_score $O/pmcw_score $O/pmcf_attn $O/pmcw_attn $O/pmc1 $O/pmc2 $O/pmcf_flash $O/pmc_flash > $O/pmc_summary.json 2>&1
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
  head -c 600 $O/pmc_summary.json
fi
if [ $part = e2e -o $part = all ]; then
  python tools/e2e_c2.py --json $O/e2e_c2.json > $O/e2e_c2.log 2>&1; echo "e2e rc=$?"
  python tools/e2e_c2.py --fused-forward --no-oracle --json $O/e2e_c2_fused_forward.json > $O/e2e_c2_fused.log 2>&1; echo "e2e fused rc=$?"
  python tools/flash2_probe.py > $O/flash_probe.txt 2>&1; echo "flash probe rc=$?"
  python tools/host_profile.py 3 > $O/host_profile.txt 2>&1
  python tools/attn_probe.py > $O/attn_probe.txt 2>&1; echo "attn probe rc=$?"
fi
