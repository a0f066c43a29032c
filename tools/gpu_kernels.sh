#!/bin/bash
# kernel unit tests + microbenchmarks
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x > gpurun_out/kernels.log 2>&1; tail -15 gpurun_out/kernels.log
timeout 600 python tools/kbench.py > gpurun_out/kbench.log 2>&1; grep -v "^parts" gpurun_out/kbench.log
JLM_GATE_TILE=64 timeout 600 python tools/kbench.py gate 2>&1 | grep split
