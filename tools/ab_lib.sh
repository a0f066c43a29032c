#!/bin/bash
# A/B two builds of libjlm_hip.so (A = in-tree, B = build_prof/libjlm_hip_b.so) on kbench filters,
# interleaved so that clock drift between runs does not decide: tools/ab_lib.sh "lse" 3
mkdir -p gpurun_out
FLT=${1:-lse}; N=${2:-3}; PAT=${3:-split}
for i in $(seq $N); do
  echo "A:"; timeout 300 python tools/kbench.py $FLT 2>&1 | grep "$PAT"
  echo "B:"; JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_b.so timeout 300 python tools/kbench.py $FLT 2>&1 | grep "$PAT"
done
