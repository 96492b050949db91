#!/bin/bash
# Round-5 measurements on the GPU box: bash tools/final_profile_r5.sh <part> ; outputs under gpurun_out/r5f/
#   bench  all BASELINE configs (driver-style command lines)      prof  rocprofv3 kernel stats (1 and 3 streams)
#   pmc    FETCH / WRITE / SQ counter passes (scoring, decode)    probes  power / co-execution probes, decode cold probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
export TMPDIR=/tmp
part=${1:-bench}
if [ $part = bench ]; then
  python bench.py > $O/bench_c4_default.json 2> $O/bench_c4_default.err; echo "c4 default rc=$?" > $O/rc.txt
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?" >> $O/rc.txt
  python bench.py --steps 5 --warmup 2 --ctx 32768 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.txt
  python bench.py --steps 3 --warmup 1 --model llama3.1-8b --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.txt
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
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc3 -o p3 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $O/pmc4 -o p4 --output-format csv -- python $R/tools/prof_score.py attn 3 > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmcf_score $O/pmcw_score $O/pmcf_attn $O/pmcw_attn $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/pmc_summary.json 2>&1
  # kernel durations of the counter runs (for the effective clock: cycles / duration)
  python - <<PY > $O/pmc_durations.txt 2>&1
import csv, glob, re, collections
for d in ("pmc1", "pmc3"):
    acc = collections.defaultdict(list)
    for path in glob.glob("$O/" + d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            m = re.search(r"(score_rowstat\d*|score_colmax\d*)", row.get("Kernel_Name", ""))
            if m: acc[m.group(1)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items(): print(d, k, "launches", len(v), "mean duration ns", sum(v) / len(v))
PY
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
  head -c 1500 $O/pmc_summary.json; cat $O/pmc_durations.txt
fi
if [ $part = probes ]; then
  mkdir -p tools/bin
  for p in probe_pipe probe_coexec; do [ -x tools/bin/$p ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/$p.hip -o tools/bin/$p 2> /dev/null; done
  for s in 0 12345; do tools/bin/probe_pipe $s > $O/probe_pipe_seed$s.txt 2>&1; done
  tools/bin/probe_coexec > $O/probe_coexec.txt 2>&1
  python tools/zero_score_probe.py > $O/zero_score_probe.txt 2>&1
  python tools/decode_cold_probe.py 205 252 > $O/decode_cold_probe.txt 2>&1
  python tools/flash2_probe.py > $O/flash2_probe.txt 2>&1
  tail -5 $O/zero_score_probe.txt $O/decode_cold_probe.txt; tail -12 $O/flash2_probe.txt
fi
