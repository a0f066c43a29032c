#!/bin/bash
# the W-stationary persistent LSTM step against the one-tile-per-workgroup kernel: unit tests, kbench A/B (JLM_GATE_V=1 | 2, ring depth, tile
# map), per-workgroup timeline of a -DJLM_PROFILE build
mkdir -p gpurun_out build_prof
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "lstm_step_xg" > gpurun_out/gate_ws_tests.log 2>&1; tail -3 gpurun_out/gate_ws_tests.log
for v in ${GATE_VARIANTS:-"JLM_GATE_V=1" "JLM_GATE_V=2" "JLM_GATE_V=2,JLM_GATE_WS_CX=2" "JLM_GATE_V=2,JLM_GATE_WS_CX=4" "JLM_GATE_V=2,JLM_GATE_WS_CX=8" "JLM_GATE_V=2,JLM_GATE_WS_CX=16" "JLM_GATE_V=1" "JLM_GATE_V=2"}; do
  echo "== $v"; env ${v//,/ } timeout 200 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done | tee gpurun_out/gate_ws_kbench.txt
if [ -n "$GATE_PROFILE" ]; then
  # build_prof/libjlm_hip_prof_abl0.so: the -DJLM_PROFILE build of the library, made on the build host (tools/probes/gate_ws_ablate.sh says how)
  for r in $GATE_PROFILE; do JLM_PROF_LIB=libjlm_hip_prof_abl0.so timeout 120 python tools/probes/gate_ws_profile.py $r; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gate_ws_timeline.txt
fi
