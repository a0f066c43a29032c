#!/bin/bash
# the LSTM step: unit tests + microbenchmark of the one-tile-per-CU table form (jlm_lstm_step_xg) beside the tile form
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "lstm_step" > gpurun_out/gate_tests.log 2>&1; tail -5 gpurun_out/gate_tests.log
for v in "0 0" "0 1" "1 0" "1 1"; do
  set -- $v
  echo "== JLM_GATE_PIPE=$1 JLM_GATE_TOUCH=$2"
  JLM_GATE_PIPE=$1 JLM_GATE_TOUCH=$2 timeout 600 python tools/kbench.py gate 2>&1 | grep "xg\|table"
done | tee gpurun_out/kbench_gate.log
