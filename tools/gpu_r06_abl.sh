#!/bin/bash
# round 6: the mx6 vocabulary kernel against its ablation builds (build_prof/libjlm_hip_<TAG>.so from tools/build_variant.sh TAG "-DMX_ABL=n" jlm_mx6.hip:
# 1 no fold, 4 no LDS-DMA in the loop, 8 no barrier, 32 no fragment reads in the loop; NS: -fno-slp-vectorize), kbench lines, interleaved twice;
# then random vs all-zero operands (the clock the governor grants)
mkdir -p gpurun_out
export KBENCH_MX6=1 KBENCH_FMTS=mx6 KBENCH_NO_BIG=1
for i in 1 2; do
  echo "== base"; timeout 300 python tools/kbench.py 2>&1 | grep "vocab_lse_mixed"
  for f in build_prof/libjlm_hip_[A-Z]*.so; do
    echo "== $(basename $f)"; JLM_HIP_LIB=$PWD/$f timeout 300 python tools/kbench.py 2>&1 | grep "vocab_lse_mixed"
  done
done
echo "== base, all-zero operands"; KBENCH_ZERO=1 timeout 300 python tools/kbench.py 2>&1 | grep "vocab_lse_mixed"
