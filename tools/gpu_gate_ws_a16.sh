#!/bin/bash
# round 6: the W-stationary LSTM step with four epilogue operand sets in flight (in-tree) against two (build_prof/libjlm_hip_WSOPS2.so:
# tools/build_variant.sh WSOPS2 "-DWS_OPS4=0" jlm_gate_ws.hip) and against every operand from one line (WSA16: -DWS_ABL=16, the bound);
# every launch size forced onto the kernel; word ids uniform over a 410-MB table and drawn ~ 1 / rank
mkdir -p gpurun_out
JLM_GATE_V=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step_xg and not forced" > gpurun_out/gate_ws_ops_tests.log 2>&1; tail -3 gpurun_out/gate_ws_ops_tests.log
{
for i in 1 2; do
for wd in uniform zipf; do
echo "== JLM_GATE_V=2 in-tree (four sets), $wd"; KBENCH_WORDS=$wd JLM_GATE_V=2 timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
echo "== JLM_GATE_V=2 WSOPS2 (two sets), $wd"; KBENCH_WORDS=$wd JLM_GATE_V=2 JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_WSOPS2.so timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done
done
echo "== JLM_GATE_V=2 WSA16 (operands from one line)"; JLM_GATE_V=2 JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_WSA16.so timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
echo "== default dispatch, uniform"; timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
echo "== default dispatch, zipf"; KBENCH_WORDS=zipf timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
} 2>&1 | tee gpurun_out/gate_ws_ops.txt
