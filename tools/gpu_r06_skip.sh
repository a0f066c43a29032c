#!/bin/bash
# round 6: the mx6 vocabulary kernel, fold-free issue slots at the head of a block (MX6_SKIP = 0 / 1 / 2 (in-tree) / 3), kbench lines, interleaved twice
mkdir -p gpurun_out
export KBENCH_MX6=1 KBENCH_FMTS=mx6 KBENCH_NO_BIG=1
for i in 1 2; do
  echo "== base (in-tree)"; timeout 300 python tools/kbench.py 2>&1 | grep "vocab_lse_mixed"
  for f in build_prof/libjlm_hip_SK*.so; do
    echo "== $(basename $f)"; JLM_HIP_LIB=$PWD/$f timeout 300 python tools/kbench.py 2>&1 | grep "vocab_lse_mixed"
  done
done
