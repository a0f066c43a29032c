#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short > gpurun_out/kernels.log 2>&1; tail -12 gpurun_out/kernels.log
timeout 600 python tools/kbench.py > gpurun_out/kbench.log 2>&1; cat gpurun_out/kbench.log
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_eval.py -m gpu -q --tb=short > gpurun_out/decode.log 2>&1; tail -8 gpurun_out/decode.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
