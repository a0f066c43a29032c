#!/bin/bash
# see host_soak8.py; build_prof/libjlm_hip_skip.so = hipcc ... -DJLM_PROBE_SKIP (tools/probes/skip_kernel.sh)
cp jlm_amd/csrc/libjlm_hip.so /tmp/libjlm_hip.keep && cp build_prof/libjlm_hip_skip.so jlm_amd/csrc/libjlm_hip.so
for n in 1 8; do python tools/probes/host_soak8.py $n ${1:-12}; done
cp /tmp/libjlm_hip.keep jlm_amd/csrc/libjlm_hip.so
