#!/bin/bash
mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/kbench.py lse 2>&1 | grep stationary
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/$tag -o p -- python $R/tools/kbench.py lse > $R/gpurun_out/pmc2/$tag.log 2>&1
  echo "$tag rc=$?"
done
