#!/bin/bash
# rocprofv3 kernel stats of an arbitrary bench configuration: tools/gpu_prof_cfg.sh <tag> <bench args...>
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-220
python3 - "$R/gpurun_out/prof_$tag/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-86s calls %6s avg %9.1f us  %5s%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
