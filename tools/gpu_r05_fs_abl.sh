#!/bin/bash
# ablations of the format-sliced kernel (wrong numbers, timing only): F1 no fold, F2 no exchange, F4 no barrier, F6, F7
for t in base F1 F2 F4 F6 F7; do
  if [ $t = base ]; then unset JLM_HIP_LIB; else export JLM_HIP_LIB=$GRAFT_REPO_ROOT/build_prof/libjlm_hip_$t.so; fi
  JLM_MX_FS=1 KBENCH_ONLY=mixed timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed .*dsoftmax" | sed "s/^/$t /"
done
