#!/bin/bash
# kbench lines for the in-tree library and every build_prof/libjlm_hip_<TAG>.so, interleaved N times
FLT=${1:-lse}; N=${2:-2}; PAT=${3:-split}
for i in $(seq $N); do
  echo "base:"; timeout 300 python tools/kbench.py $FLT 2>&1 | grep "$PAT"
  for f in build_prof/libjlm_hip_[A-Z]*.so; do
    echo "$(basename $f):"; JLM_HIP_LIB=$PWD/$f timeout 300 python tools/kbench.py $FLT 2>&1 | grep "$PAT"
  done
done
