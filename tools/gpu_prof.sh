#!/bin/bash
# rocprofv3 kernel stats of the default bench (the command the driver runs, shorter)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:---no-config5} > $R/gpurun_out/$TAG.log 2>&1
echo "prof rc=$?"; tail -1 $R/gpurun_out/$TAG.log | cut -c1-400
python - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/$TAG/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print("%-100s calls %6s avg %9.1f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
