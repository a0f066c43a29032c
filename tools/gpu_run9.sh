#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_eval.py -m gpu -q --tb=short -x > gpurun_out/decode.log 2>&1; tail -8 gpurun_out/decode.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
