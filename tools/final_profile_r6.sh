#!/bin/bash
# Round 6, the final default path (pruned scoring call with candidates at key granularity, both dtypes, three side streams): rocprofv3 kernel
# stats of the bench line (1 / 3 streams), a kernel-trace timeline of the scoring loop, and FETCH / WRITE / SQ counter passes of the scoring call.
#   bash tools/final_profile_r6.sh ; outputs under gpurun_out/r6p/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof3 -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 8 > $O/prof3_bench.json 2> $O/prof3.err
rocprofv3 --kernel-trace --stats -d $O/prof1 -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 8 --score-streams 1 > $O/prof1_bench.json 2> $O/prof1.err
python $R/tools/timeline_score.py $(find $O/prof3 -name "*kernel_trace.csv" | head -1) > $O/timeline_3streams.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf -o f --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcw -o w --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc1 -o p1 --output-format csv -- python $R/tools/prof_score.py score 3 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmcf $O/pmcw $O/pmc1 > $O/pmc_summary.json 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
ls $O/prof3 $O/prof1; head -c 3000 $O/pmc_summary.json; cat $O/timeline_3streams.txt
