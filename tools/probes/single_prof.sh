#!/bin/bash
# per-kernel averages of sentence-at-a-time decodes (tied V=50k, 20 kana, beam 10)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_single -o run -- python $R/tools/probes/single_sentence.py mid-tied > $R/gpurun_out/prof_single.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_single/run_kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s calls %6s  avg %8.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
