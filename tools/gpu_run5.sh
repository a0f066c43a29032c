#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lse or gemm or lstm" > gpurun_out/kernels.log 2>&1; tail -4 gpurun_out/kernels.log
timeout 600 python tools/kbench.py > gpurun_out/kbench.log 2>&1; cat gpurun_out/kbench.log
