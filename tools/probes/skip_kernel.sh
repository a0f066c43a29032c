#!/bin/bash
# What is on the critical path of the pipelined step: the device-resident loop with single kernels left out of the frame loop
# (-DJLM_PROBE_SKIP build of the library, swapped in for this run only; results are wrong by construction).  Build first:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_PROBE_SKIP -Iinclude -Ijlm_amd/csrc \
#     -o build_prof/libjlm_hip_skip.so jlm_amd/csrc/jlm_{gemm,beam,split,gate,decode}.hip
cp jlm_amd/csrc/libjlm_hip.so /tmp/libjlm_hip.keep && cp build_prof/libjlm_hip_skip.so jlm_amd/csrc/libjlm_hip.so
for s in 0 1 2 4 8 16 7 15 0; do
  echo "JLM_SKIP=$s: $(JLM_SKIP=$s timeout 200 python tools/ab_streams.py ${CFG:-3,3,66} 2>&1 | tail -1 | cut -c1-110)"
done
cp /tmp/libjlm_hip.keep jlm_amd/csrc/libjlm_hip.so
