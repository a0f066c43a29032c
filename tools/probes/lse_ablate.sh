#!/bin/bash
# Time and shader clock of the vocabulary kernel with parts of it switched off (-DJLM_ABL bits: 1 fold, 2 MFMAs, 4 fragment reads,
# 8 LDS-DMA).  Build on the CPU side first:
#   for a in 0 1 2 4 8 5 12 13; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_WGTIME -DJLM_ABL=$a -Iinclude \
#     -Ijlm_amd/csrc -o build_prof/libjlm_hip_abl$a.so jlm_amd/csrc/jlm_{gemm,beam,split,gate,decode}.hip; done
for a in ${ABLS:-0 1 2 4 8 5 12 13}; do
  echo "== JLM_ABL=$a"
  JLM_PROF_LIB=libjlm_hip_abl$a.so timeout 120 python tools/probes/lse_wg_timeline.py 2>&1 | grep -E "kernel span|shader clock"
done
