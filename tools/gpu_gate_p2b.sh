#!/bin/bash
# round 6: the 128 x 256 LSTM step (JLM_GATE_V=4) against the default dispatch in the microbenchmark, word ids uniform over a 410-MB table
# and drawn ~ 1 / rank; P2_ABL builds if present (8: every epilogue operand from one line)
mkdir -p gpurun_out
JLM_GATE_V=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step_xg" > gpurun_out/gate_p2_tests.log 2>&1; tail -3 gpurun_out/gate_p2_tests.log
{
for i in 1 2; do
for wd in uniform zipf; do
  echo "== default dispatch, words $wd"; KBENCH_WORDS=$wd timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
  echo "== JLM_GATE_V=4, words $wd"; KBENCH_WORDS=$wd JLM_GATE_V=4 timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done
done
for f in build_prof/libjlm_hip_P2*.so; do
  [ -f $f ] || continue
  echo "== JLM_GATE_V=4 $(basename $f)"; JLM_GATE_V=4 JLM_HIP_LIB=$PWD/$f timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done
} 2>&1 | tee gpurun_out/gate_p2b_kbench.txt
