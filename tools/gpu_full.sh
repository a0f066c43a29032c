#!/bin/bash
# full validation + measurement pass: GPU tests, smoke, bench (with CPU baseline), rocprof stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_tests.log 2>&1; tail -5 gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-1500
bash tools/gpu_prof.sh prof
