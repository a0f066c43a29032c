#!/bin/bash
# SQ counters of the LSTM-step kernel (kbench "gate" workloads), one rocprofv3 --pmc pass per counter set
mkdir -p gpurun_out/pmc_gate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export JLM_GATE_TOUCH=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gate/$tag -o p -- python $R/tools/kbench.py gate > $R/gpurun_out/pmc_gate/$tag.log 2>&1
  echo "$tag rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_gate/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(R + "/gpurun_out/pmc_gate/summary.txt", "w") as out:
    for k in sorted(agg):
        if "gate" not in k and "gemm_split" not in k: continue
        print(k); out.write(k + "\n")
        for c, v in sorted(agg[k].items()):
            ln = "   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v))
            print(ln); out.write(ln + "\n")
PY
