#!/bin/bash
# SQ counters of the headline decode's kernels (the kernels the bench line is about, in the forms the loader picked -- e.g. the mx6 normaliser
# without a running maximum, which kbench's fixed scales do not reach): one rocprofv3 --pmc pass per counter set over a short bench run
TAG=${1:-pmcb}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/$tag -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-legs > $R/gpurun_out/$TAG/$tag.log 2>&1
  echo "$tag rc=$?"
done
python - <<PY
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not any(x in k for x in ("vocab_lse", "gate_xg", "gemm_split3", "beam_step", "wordlist_kernel", "pack_t")): continue
    n = max(len(v) for v in agg[k].values())
    if n < 20: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
