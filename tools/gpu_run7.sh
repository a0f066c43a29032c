#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lse or gemm or lstm" > gpurun_out/kernels.log 2>&1; tail -6 gpurun_out/kernels.log
for v2 in 0 1; do echo "== V2=$v2"; JLM_GEMM_V2=$v2 timeout 300 python tools/kbench.py variants 2>&1 | grep -v amdgpu.ids; JLM_GEMM_V2=$v2 timeout 300 python tools/kbench.py 2>&1 | grep "lstm_step\|vocab_lse \|gemm_nt"; done > gpurun_out/v2.log 2>&1
cat gpurun_out/v2.log
