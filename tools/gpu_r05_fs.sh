#!/bin/bash
# round 5 experiment: the format-sliced wave pairs (jlm_mixed_fs.hip, JLM_MX_FS=1) against the eight-wave kernel: kernel tests, bit-for-bit
# comparison of the partial slices, launch times
mkdir -p gpurun_out
JLM_MX_FS=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lse_mixed" --tb=short 2>&1 | tail -5
for fs in 0 1; do
  JLM_MX_FS=$fs KBENCH_ONLY=mixed KBENCH_DUMP=/tmp/fs$fs timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_mixed .*dsoftmax\|Error\|error" | sed "s/^/FS=$fs /"
done
python - <<'P'
import numpy as np, glob
for f0 in sorted(glob.glob("/tmp/fs0.*.npy")):
    f1 = f0.replace("/tmp/fs0.", "/tmp/fs1.")
    a, b = np.load(f0), np.load(f1)
    same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print(f0.split("fs0.")[1], a.shape, "bit-identical" if same else "DIFFER: max |d| %.3g, %d of %d values" % (np.abs(a - b).max() if a.shape == b.shape else -1, int((a != b).sum()) if a.shape == b.shape else -1, a.size))
P
