#!/bin/bash
# round 6: SQ counters of the mx6 vocabulary kernels (kbench KBENCH_MX6 lines), one rocprofv3 --pmc pass per counter set; env passes through
# (JLM_MX6_PAIR, JLM_MX6_WIDE, KBENCH_FMTS).  usage: bash tools/gpu_pmc_mx6.sh <tag>
TAG=${1:-pmc6}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export KBENCH_MX6=1 KBENCH_NO_BIG=1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/$tag -o p -- python $R/tools/kbench.py > $R/gpurun_out/$TAG/$tag.log 2>&1
  echo "$tag rc=$?"
done
python - <<PY
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:64] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "lse" not in k: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
