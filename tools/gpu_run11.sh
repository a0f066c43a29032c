#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q --tb=short -x > gpurun_out/decode.log 2>&1; tail -3 gpurun_out/decode.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','end_to_end_chars_per_s')}, d['roofline']['frac'], d['gate_gemm']['mfma_util_pct'])"
