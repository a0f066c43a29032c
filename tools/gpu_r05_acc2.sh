#!/bin/bash
# round 5: the two-accumulator-pair body of the mixed vocabulary kernel (in-tree: k <= 100 segments; A7: also k = 200; A0: none)
mkdir -p gpurun_out
O=gpurun_out/r05_a_acc2.txt; : > $O
for lib in "" A7; do
  echo "== tests lib=${lib:-intree}" >> $O
  JLM_HIP_LIB=${lib:+$PWD/build_prof/libjlm_hip_$lib.so} timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mixed_logits.py -m gpu -q --tb=short -x -k "mixed or hybrid" 2>&1 | tail -3 >> $O
done
for lib in A0 A7; do
  KBENCH_DUMP=gpurun_out/dump_$lib KBENCH_ONLY=mixed KBENCH_SEGS=1 JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_$lib.so timeout 300 python tools/kbench.py lse > /dev/null 2>&1
done
KBENCH_DUMP=gpurun_out/dump_A4 KBENCH_ONLY=mixed KBENCH_SEGS=1 timeout 300 python tools/kbench.py lse > /dev/null 2>&1
python - >> $O <<'P'
import glob, numpy as np
for f in sorted(glob.glob("gpurun_out/dump_A0.*.npy")):
    a = np.load(f)
    for t in ("A4", "A7"):
        b = np.load(f.replace("dump_A0", "dump_" + t))
        print("bitcmp", f.split("dump_A0.")[1], t, "identical" if a.tobytes() == b.tobytes() else "DIFF max %g" % np.abs(a - b).max())
P
rm -f gpurun_out/dump_*.npy
for i in 1 2 3; do
  for lib in "" A0 A7; do
    echo "lib=${lib:-intree(A4)}:" >> $O
    KBENCH_ONLY=mixed KBENCH_SEGS=1 JLM_HIP_LIB=${lib:+$PWD/build_prof/libjlm_hip_$lib.so} timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed" >> $O
  done
done
cat $O
