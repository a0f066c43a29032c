#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log
tail -8 gpurun_out/kernels.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "prof rc=$?"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20
