#!/bin/bash
# round 5: is the vocabulary kernel bound by the power governor?  The same launches on random and on all-zero operands (KBENCH_ZERO=1:
# identical instruction streams, minimal switching power), eight-wave and wide kernels and the wide kernel without its fold.
mkdir -p gpurun_out
O=gpurun_out/r05_d_power.txt; : > $O
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -4 >> $O
for i in 1 2; do
  for z in "" 1; do
    for cfg in "0:" "1:" "1:W1" "1:W33"; do
      w=${cfg%%:*}; lib=${cfg##*:}
      echo "zero=${z:-0} JLM_MX_WIDE=$w lib=${lib:-intree}:" >> $O
      KBENCH_ZERO=$z JLM_MX_WIDE=$w JLM_HIP_LIB=${lib:+$PWD/build_prof/libjlm_hip_$lib.so} KBENCH_ONLY=mixed KBENCH_SEGS=1 timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed" | grep -v tied | sed 's/V=[0-9]* //; s/([0-9.]*% of f32 MFMA peak)//; s/vocab_lse_mixed *//' >> $O
    done
  done
done
cat $O
