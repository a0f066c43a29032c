#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "stationary" > gpurun_out/kernels.log 2>&1; tail -4 gpurun_out/kernels.log
for abl in 0; do timeout 300 python tools/kbench.py lse 2>&1 | grep "stationary"; done > gpurun_out/stat.log 2>&1
cat gpurun_out/stat.log
