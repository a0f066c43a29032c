#!/bin/bash
# the vocabulary kernel's LDS-DMA instructions behind the first block's matrix instructions (A, in-tree) vs in front of them
# (B, build_prof/libjlm_hip_b.so): kernel + golden tests, then kbench interleaved (three-segment launch, segments alone, tied shapes)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mixed_logits.py -m gpu -q --tb=short -x -k "mixed or lse or hybrid" > gpurun_out/lse_spread_tests.log 2>&1; tail -3 gpurun_out/lse_spread_tests.log
for i in 1 2 3; do
  echo "A:"; KBENCH_SEGS=1 timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed\|vocab_lse_hybrid"
  echo "B:"; KBENCH_SEGS=1 JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_b.so timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed\|vocab_lse_hybrid"
done
