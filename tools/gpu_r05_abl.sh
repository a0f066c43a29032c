#!/bin/bash
# round 5: what the mixed kernel's VALU work costs it -- timing-model ablations (wrong numbers on purpose), interleaved kbench
#   A0 shipped body | B64 no running maximum, no scaling (exp2 + add per logit) | B192 = B64 + no cvt / fma combine | B1 no fold | B3 no fold, no combine
mkdir -p gpurun_out
O=gpurun_out/r05_b_valu_ablate.txt; : > $O
for i in 1 2; do
  for lib in A0 B64 B192 B1 B3; do
    echo "lib=$lib:" >> $O
    KBENCH_ONLY=mixed KBENCH_SEGS=1 JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_$lib.so timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed" | grep -v tied >> $O
  done
done
cat $O
