#!/bin/bash
# round 5: the wide form of the mixed vocabulary kernel (JLM_MX_WIDE=1: four waves x 64 rows, jlm_mixed_w.hip) against the eight-wave one
mkdir -p gpurun_out
O=gpurun_out/r05_c_wide.txt; : > $O
echo "== tests JLM_MX_WIDE=1" >> $O
JLM_MX_WIDE=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mixed_logits.py -m gpu -q --tb=short -x -k "mixed or hybrid" 2>&1 | tail -15 >> $O
for i in 1 2 3; do
  for w in 0 1; do
    echo "JLM_MX_WIDE=$w:" >> $O
    JLM_MX_WIDE=$w KBENCH_ONLY=mixed KBENCH_SEGS=1 timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed" | grep -v tied >> $O
  done
  for lib in build_prof/libjlm_hip_W*.so; do
    echo "JLM_MX_WIDE=1 $(basename $lib):" >> $O
    JLM_MX_WIDE=1 JLM_HIP_LIB=$PWD/$lib KBENCH_ONLY=mixed KBENCH_SEGS=1 timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed" | grep -v tied >> $O
  done
done
cat $O
