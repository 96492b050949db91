#!/bin/bash
# Round-6 measurements, second session (pipelined tail default): bash tools/final_profile_r6b.sh <part> ; outputs under gpurun_out/r6bp/
#   prof   rocprofv3 kernel stats of the bench line (1 / 3 side streams), a kernel-trace timeline of the scoring loop, FETCH / WRITE / SQ counter
#          passes of the scoring call (the default path: pruned call with candidates at key granularity)
#   bench  all BASELINE configs (driver-style command lines)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6bp; mkdir -p $O; export TMPDIR=/tmp
part=${1:-prof}
if [ $part = prof ]; then
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/prof3 -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 8 > $O/prof3_bench.json 2> $O/prof3.err
  rocprofv3 --kernel-trace --stats -d $O/prof1 -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 8 --score-streams 1 > $O/prof1_bench.json 2> $O/prof1.err
  python $R/tools/timeline_score.py $(find $O/prof3 -name "*kernel_trace.csv" | head -1) > $O/timeline_3streams.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf -o f --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcw -o w --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  PROF_DTYPE=bf16 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc1b -o p1b --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmcf $O/pmcw $O/pmc1 > $O/pmc_summary.json 2>&1
  python tools/pmc_summary.py $O/pmc1b > $O/pmc_summary_bf16.json 2>&1
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
  ls $O/prof3 $O/prof1; cat $O/pmc_summary.json | tr -d '\n ' | head -c 2500; echo; head -25 $O/timeline_3streams.txt
fi
if [ $part = bench ]; then
  cd $R
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4_driver_style.json 2> $O/bench_c4_driver_style.err; echo "c4 driver style rc=$?" > $O/rc.txt
  python bench.py > $O/bench_c4_default.json 2> $O/bench_c4_default.err; echo "c4 default rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --dtype bf16 > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; echo "c4bf16 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --ctx 32768 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.txt
  python bench.py --steps 3 --warmup 1 --model llama3.1-8b --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --model qwen2.5-14b --level head --dtype bf16 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --score-streams 1 --no-cpu-baseline > $O/bench_c4_1stream.json 2> $O/bench_c4_1stream.err; echo "c4 1stream rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --tune score_prune=0 --no-cpu-baseline > $O/bench_c4_two_pass.json 2> $O/bench_c4_two_pass.err; echo "c4 two-pass rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --tune score_prune=5 --no-cpu-baseline > $O/bench_c4_pair_level.json 2> $O/bench_c4_pair_level.err; echo "c4 pair-level rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --tune score_prune=3 --no-cpu-baseline > $O/bench_c4_chained_tail.json 2> $O/bench_c4_chained_tail.err; echo "c4 chained-tail rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline > $O/bench_c4_force_dist.json 2> $O/bench_c4_force_dist.err; echo "c4 force-dist rc=$?" >> $O/rc.txt
  cat $O/rc.txt
  for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],2), 'decode', d['decode']['ms_per_token'] and round(d['decode']['ms_per_token'],3), 'parity_ok', d.get('parity_ok'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
  done
fi
