#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lse" > gpurun_out/kernels.log 2>&1; tail -3 gpurun_out/kernels.log
timeout 600 python tools/kbench.py > gpurun_out/kbench.log 2>&1; cat gpurun_out/kbench.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
