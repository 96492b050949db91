#!/bin/bash
# Round-4 measurements on the GPU box: bash tools/final_profile_r4.sh <part> ; outputs under gpurun_out/r4/
#   ab     same-box A/B of the headline bench: the round-3 tree (tools/ab/r3_tree, built library included) vs this tree, interleaved
#   bench  all BASELINE configs        prof  rocprofv3 kernel stats (1 and 3 streams)      pmc  FETCH / WRITE / SQ counter passes
#   probes flash2 XCD order, decode attention probe, host profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export TMPDIR=/tmp
part=${1:-ab}
if [ $part = ab ]; then
  for i in 1 2; do
    (cd tools/ab/r3_tree && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --decode-tokens 8 > $O/ab_r3_$i.json 2> $O/ab_r3_$i.err)
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --decode-tokens 8 > $O/ab_r4_$i.json 2> $O/ab_r4_$i.err
  done
  python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        st = d["roofline_stages"]
        print(os.path.basename(f), round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms | rowstat", round(st["score_rowstat"]["avg_ms"] * 1e3, 1),
              "colmax", round(st["score_colmax"]["avg_ms"] * 1e3, 1), "select us", round(st["select"]["avg_ms"] * 1e3, 1), "frac", round(st["select"]["frac"], 3),
              "| compact", round(st["compact_gather"]["frac"], 3), "| decode ms/token", round(d["decode"]["ms_per_token"], 3), "frac", round(st["decode_varlen_attn"]["frac"], 3))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
fi
if [ $part = bench ]; then
  python bench.py --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?" > $O/rc.txt
  python bench.py --steps 5 --warmup 2 --ctx 32768 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.txt
  python bench.py --steps 3 --warmup 1 --model llama3.1-8b > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --model qwen2.5-14b --level head --dtype bf16 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err; echo "c4bf16 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --score-streams 1 --no-cpu-baseline > $O/bench_c4_1stream.json 2> $O/bench_c4_1stream.err; echo "c4 1stream rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline > $O/bench_c4_force_dist.json 2> $O/bench_c4_force_dist.err; echo "c4 force-dist rc=$?" >> $O/rc.txt
  cat $O/rc.txt
fi
if [ $part = prof ]; then
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/prof1 -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 8 --score-streams 1 > $O/prof1_bench.json 2> $O/prof1.err
  rocprofv3 --kernel-trace --stats -d $O/prof3 -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 8 > $O/prof3_bench.json 2> $O/prof3.err
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
  ls $O/prof1 $O/prof3
  cd $R
fi
if [ $part = pmc ]; then
  cd /tmp
  for what in score attn; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf_$what -o f --output-format csv -- python $R/tools/prof_score.py $what 3 > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcw_$what -o w --output-format csv -- python $R/tools/prof_score.py $what 3 > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc2 -o p2 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf_flash -o f --output-format csv -- python $R/tools/prof_score.py flash 2 > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmcf_score $O/pmcw_score $O/pmcf_attn $O/pmcw_attn $O/pmc1 $O/pmc2 $O/pmcf_flash > $O/pmc_summary.json 2>&1
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
  head -c 1200 $O/pmc_summary.json
fi
if [ $part = probes ]; then
  python tools/flash2_xcd_probe.py > $O/flash2_xcd_probe.txt 2>&1; echo "flash2 xcd probe rc=$?"; cat $O/flash2_xcd_probe.txt
  python tools/attn_probe.py > $O/attn_probe.txt 2>&1; echo "attn probe rc=$?"; tail -12 $O/attn_probe.txt
fi
