#!/bin/bash
# the persistent one-tile-per-CU LSTM step (JLM_GATE_V=3) against the round-3 kernel (1) and the W-stationary one (2): unit tests under
# each, then kbench A/B
mkdir -p gpurun_out
for v in 3 2; do
  JLM_GATE_V=$v timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "lstm_step_xg" 2>&1 | tail -3
done | tee gpurun_out/gate_pu_tests.log
for v in ${GATE_VARIANTS:-"JLM_GATE_V=1" "JLM_GATE_V=3" "JLM_GATE_V=2" "JLM_GATE_V=1" "JLM_GATE_V=3" "JLM_GATE_V=2"}; do
  echo "== $v"; env ${v//,/ } timeout 200 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done | tee gpurun_out/gate_pu_kbench.txt
