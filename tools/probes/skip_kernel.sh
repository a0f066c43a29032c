#!/bin/bash
# What is on the critical path of the pipelined step: the device-resident loop with single kernels left out of the frame loop
# (-DJLM_PROBE_SKIP build of the library, swapped in for this run only; results are wrong by construction).  Build first:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_PROBE_SKIP -Iinclude -Ijlm_amd/csrc \
#     -o build_prof/libjlm_hip_skip.so jlm_amd/csrc/jlm_{gemm,beam,split,gate,mixed,decode}.hip
# AFTER=n (default 14 = the warm-up loops of tools/ab_streams.py): the first n frame loops run complete, so that T / Tm / h keep data of
# the usual kind -- zeros in the matrix kernels' operands draw less power, the clock rises and the probe overstates what a kernel costs.
# Bits: 1 edge logits, 2 T projection, 4 LSTM step, 8 beam step (degenerate: no live rows), 16 vocabulary kernel, 32 packing of the T rows.
cp jlm_amd/csrc/libjlm_hip.so /tmp/libjlm_hip.keep && cp build_prof/libjlm_hip_skip.so jlm_amd/csrc/libjlm_hip.so
for s in ${SKIPS:-0 1 2 32 34 4 16 0}; do
  echo "JLM_SKIP=$s: $(JLM_SKIP_AFTER=${AFTER:-14} JLM_SKIP=$s timeout 200 python tools/ab_streams.py ${CFG:-4,4,66} 2>&1 | tail -1 | cut -c1-110)"
done
cp /tmp/libjlm_hip.keep jlm_amd/csrc/libjlm_hip.so
