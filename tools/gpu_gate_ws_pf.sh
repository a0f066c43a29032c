#!/bin/bash
# round 6: the W-stationary LSTM step with the epilogue's lines warmed at k-step WS_PF (in-tree: 8) against -DWS_PF=-1 / 12 builds
# (build_prof/libjlm_hip_PFOFF.so, _PF12.so: tools/build_variant.sh PFOFF "-DWS_PF=-1" jlm_gate_ws.hip), every launch size forced onto it
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step_xg" > gpurun_out/gate_ws_pf_tests.log 2>&1; tail -2 gpurun_out/gate_ws_pf_tests.log
JLM_GATE_V=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step_xg" > gpurun_out/gate_ws_pf_tests_v2.log 2>&1; tail -2 gpurun_out/gate_ws_pf_tests_v2.log
{
echo "== default dispatch (u16 / pu / ws by rows), in-tree"; timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
for i in 1 2; do
  echo "== JLM_GATE_V=2 in-tree (WS_PF=8)"; JLM_GATE_V=2 timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
  for f in build_prof/libjlm_hip_PF*.so; do
    echo "== JLM_GATE_V=2 $(basename $f)"; JLM_GATE_V=2 JLM_HIP_LIB=$PWD/$f timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
  done
done
} | tee gpurun_out/gate_ws_pf_kbench.txt
