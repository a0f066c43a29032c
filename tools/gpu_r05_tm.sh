#!/bin/bash
# round 5: granule-major packed hypothesis rows (coalesced operand loads): kernel tests for both forms of the mixed kernel, kbench
mkdir -p gpurun_out
O=gpurun_out/r05_g_tm_layout.txt; : > $O
for w in 0 1; do
echo "== tests JLM_MX_WIDE=$w" >> $O
JLM_MX_WIDE=$w timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mixed_logits.py -m gpu -q --tb=short -x -k "mixed or hybrid" 2>&1 | tail -15 >> $O
done
for i in 1; do
  for w in 0 1; do
    echo "JLM_MX_WIDE=$w:" >> $O
    JLM_MX_WIDE=$w KBENCH_ONLY=mixed KBENCH_SEGS=1 timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed\|pack_t_mixed" | sed 's/V=[0-9]* //; s/([ 0-9.]*% of f32 MFMA peak)//; s/vocab_lse_mixed *//' >> $O
  done
done
cat $O
