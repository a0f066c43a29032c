#!/bin/bash
# PMC counters for the GEMM kernels (own run: --pmc with --kernel-trace only)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
grep -c . $R/gpurun_out/pmc/counters.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tools/kbench.py "$1" > $R/gpurun_out/pmc/$tag.log 2>&1
  echo "$tag rc=$?"
done
ls -R $R/gpurun_out/pmc | head -40
