#!/bin/bash
# HBM traffic of the dominant kernels: FETCH_SIZE and WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
mkdir -p gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-legs > $R/gpurun_out/traffic/$c.log 2>&1
  echo "$c rc=$?"
done
