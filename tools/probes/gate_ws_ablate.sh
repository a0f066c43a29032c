#!/bin/bash
# the W-stationary LSTM step with parts of its k-steps switched off (-DWS_ABL builds, made on the build host: results are wrong, the time is
# the answer): per-workgroup timelines.  usage: gate_ws_ablate.sh "0 1 2 3 8" "2560 20480"
#   for abl in ...; hipcc ... -DJLM_PROFILE -DWS_ABL=$abl -mllvm -amdgpu-mfma-vgpr-form -c -o build_prof/ws_abl$abl.o jlm_gate_ws.hip; hipcc -shared -o build_prof/libjlm_hip_prof_abl$abl.so build_prof/jlm_*.o build_prof/ws_abl$abl.o
mkdir -p gpurun_out
for abl in $1; do
  echo "=== WS_ABL=$abl (1 no LDS-DMA in the k-steps, 2 no fragment reads, 8 no barrier)"
  for r in ${2:-20480}; do JLM_PROF_LIB=libjlm_hip_prof_abl$abl.so timeout 120 python tools/probes/gate_ws_profile.py $r 2>&1 | grep -v "^  tile [2-6]\|amdgpu.ids"; done
done | tee gpurun_out/gate_ws_ablate.txt
