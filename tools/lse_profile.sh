#!/bin/bash
# per-wave cycle accounting of the vocabulary LSE kernel; the library is built on the CPU side:
#   mkdir -p build_prof && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_PROFILE \
#       -o build_prof/libjlm_hip_prof.so jlm_amd/csrc/jlm_gemm.hip jlm_amd/csrc/jlm_beam.hip
mkdir -p gpurun_out
timeout 600 python tools/lse_profile.py > gpurun_out/lse_profile.log 2>&1
cat gpurun_out/lse_profile.log
