#!/bin/bash
# jlm_vocab_lse_mixed per segment (kbench) for the in-tree build and the -DMX_ABL=<bits> builds in build_prof/
# (1 no in-stream fold, 2 no combine, 4 no DMA in the loop, 8 no barrier; see csrc/jlm_mixed_body.h)
echo base; KBENCH_SEGS=1 KBENCH_ONLY=seg timeout 300 python tools/kbench.py lse 2>&1 | grep "mixed   \|split+bcol"
for f in build_prof/libjlm_hip_ABL*.so; do echo "$f"; JLM_HIP_LIB=$PWD/$f KBENCH_SEGS=1 KBENCH_ONLY=seg timeout 300 python tools/kbench.py lse 2>&1 | grep "mixed   "; done
