#!/bin/bash
# full validation + measurement pass: GPU tests, smoke, bench (with CPU baseline), rocprof stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_tests.log 2>&1; tail -5 gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
echo "prof rc=$?"; ls $R/gpurun_out/prof | head
