#!/bin/bash
# rows-stationary LSE kernels: unit tests (f32 and split-f16 forms) + microbenchmarks
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -s -k "stationary or split" > gpurun_out/kernels_lse.log 2>&1; grep -E "max \|lse|passed|failed|Error|error" gpurun_out/kernels_lse.log | head -40
timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_s" > gpurun_out/stat_lse.log
cat gpurun_out/stat_lse.log
