#!/bin/bash
# per-tile timeline of the 128 x 256 persistent LSTM step (build_prof/libjlm_hip_PROF.so = -DJLM_PROFILE build; _PROFA3.so: the same without
# LDS-DMA and fragment reads in the k-steps), and of the persistent 160 x 128 kernel beside it if its profile build is there
mkdir -p gpurun_out
{
for r in ${1:-10240 20480}; do
  JLM_PROF_LIB=libjlm_hip_PROF.so timeout 120 python tools/probes/gate_p2_profile.py $r
  echo "-- no LDS-DMA, no fragment reads in the k-steps (P2_ABL=3)"
  JLM_PROF_LIB=libjlm_hip_PROFA3.so timeout 120 python tools/probes/gate_p2_profile.py $r
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gate_p2_timeline.txt
