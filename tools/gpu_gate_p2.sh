#!/bin/bash
# round 6: the 128 x 256 / 2 x 2-register-block persistent LSTM step (csrc/jlm_gate_p2.hip, JLM_GATE_V=4) -- unit tests with every launch forced onto it,
# kbench lines against the default dispatch (one tile per workgroup / persistent 160 x 128 / W-stationary), ablation builds if present
mkdir -p gpurun_out
JLM_GATE_V=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step_xg" > gpurun_out/gate_p2_tests.log 2>&1; tail -15 gpurun_out/gate_p2_tests.log
{
for i in 1 2; do
  echo "== default dispatch"; timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
  echo "== JLM_GATE_V=4"; JLM_GATE_V=4 timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done
for f in build_prof/libjlm_hip_P2*.so; do
  [ -f $f ] || continue
  echo "== JLM_GATE_V=4 $(basename $f)"; JLM_GATE_V=4 JLM_HIP_LIB=$PWD/$f timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done
} 2>&1 | tee gpurun_out/gate_p2_kbench.txt
