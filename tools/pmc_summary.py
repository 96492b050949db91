#!/usr/bin/env python
"""Average per-launch counter values per kernel from rocprofv3 --pmc output directories (csv):
   python tools/pmc_summary.py dir [dir ...]  ->  JSON {kernel: {counter: average}}  (kernel names shortened)"""
import csv, glob, json, os, re, sys
out = {}
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                m = re.search(r"(score_rowstatT2|score_tail|score_colmax_sparse|score_colmax_keys|score_merge|score_bounds2|score_bounds3|score_rowstat\d*|score_colmax\d*|flash2?_fwd|varlen_attn_split\d*|varlen_attn_combine\d*|compact_gather\w*|dense_append|score_finalize\w*)", name)
                if not m:
                    continue
                key = (m.group(1), row["Counter_Name"])
                v = float(row["Counter_Value"])
                s, n = acc.get(key, (0.0, 0))
                acc[key] = (s + v, n + 1)
        # one CSV row per (dispatch, counter) - possibly per XCD/SE instance summed by the tool already
        for (k, c), (s, n) in acc.items():
            # launches = rows / instances is unknown here: report the per-row mean and the row count
            out.setdefault(k, {})[c] = {"mean": s / n, "rows": n, "sum": s}
print(json.dumps(out, indent=1, sort_keys=True))
