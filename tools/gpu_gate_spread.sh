#!/bin/bash
# the k-step's LDS-DMA instructions one behind each MFMA (A, in-tree) vs all behind the first (B, build_prof/libjlm_hip_b.so):
# kernel tests, then kbench interleaved for the three launch forms
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "lstm_step" > gpurun_out/gate_spread_tests.log 2>&1; tail -3 gpurun_out/gate_spread_tests.log
for i in 1 2 3; do
  for v in 1 3 2; do
    echo "A V=$v:"; JLM_GATE_V=$v timeout 200 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
    echo "B V=$v:"; JLM_GATE_V=$v JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_b.so timeout 200 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
  done
done
