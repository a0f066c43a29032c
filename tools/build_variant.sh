#!/bin/bash
# Build build_prof/libjlm_hip_<TAG>.so: the in-tree objects, with the listed sources recompiled under extra flags.
#   tools/build_variant.sh A4 "-DMX_ACC2_MAX_NB=4" jlm_mixed.hip jlm_split.hip
# (A/B partner of tools/ab_libs.sh; python __graft_entry__.py must have built the in-tree objects first)
set -e
TAG=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../jlm_amd/csrc"
mkdir -p ../../build_prof/$TAG
OBJS=""
for s in jlm_gemm jlm_beam jlm_split jlm_gate jlm_gate_p2 jlm_mixed jlm_decode jlm_gate_ws jlm_mixed_w jlm_mx6 jlm_mx6w; do
  if [[ " $* " == *" $s.hip "* ]]; then
    X=""; { [ $s = jlm_gate_ws ] || [ $s = jlm_mixed_w ]; } && X="-mllvm -amdgpu-mfma-vgpr-form"
    [ $s = jlm_mx6 ] && X="-fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize"
    [ $s = jlm_mx6w ] && X="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -mno-amdgpu-ieee"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS $X -c -o ../../build_prof/$TAG/$s.o $s.hip &
    OBJS="$OBJS ../../build_prof/$TAG/$s.o"
  else
    OBJS="$OBJS $s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../build_prof/libjlm_hip_$TAG.so $OBJS
echo built build_prof/libjlm_hip_$TAG.so
