#!/bin/bash
# round 6: the generation step as a Python loop vs a replayed HIP graph, kernel by kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6dec; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in loop graph; do
  MODE=$m python $R/tools/decode_graph_probe.py
  MODE=$m rocprofv3 --kernel-trace -d $O/$m -o t --output-format csv -- python $R/tools/decode_graph_probe.py > $O/$m.txt 2>&1
  echo "== $m (under rocprofv3)"; tail -1 $O/$m.txt; python $R/tools/decode_trace_summary.py $(find $O/$m -name "*kernel_trace.csv" | head -1)
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" -delete
