#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dyn -o run -- python $R/bench.py --fixture mid-tied --decoder dynamic --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_dyn.log 2>&1
head -9 $R/gpurun_out/prof_dyn/run_kernel_stats.csv | cut -c1-70,160-300
