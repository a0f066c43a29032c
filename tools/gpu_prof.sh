#!/bin/bash
# rocprofv3 kernel stats of the default bench + graph-replay A/B
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
JLM_GRAPH=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log | cut -c1-260
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_nograph.log 2>&1; tail -1 gpurun_out/bench_nograph.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
echo "prof rc=$?"; ls $R/gpurun_out/prof | head
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-90s calls %6s avg %9.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
