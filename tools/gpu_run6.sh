#!/bin/bash
mkdir -p gpurun_out
for v in 0 6 7 8 9 10; do JLM_GEMM_VARIANT=$v timeout 300 python tools/kbench.py variants 2>&1 | grep -v amdgpu.ids; done > gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log
