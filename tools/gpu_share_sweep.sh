#!/bin/bash
# CU share of the vocabulary kernel in the pipelined decode (JLM_LSE_SHARE, percent) x batches in flight (JLM_STREAMS): device-resident
# and strings -> strings ms per step at BASELINE configs[1]
mkdir -p gpurun_out
for st in 4 3; do for sh in 50 58 66 75 83 100 66; do
  JLM_STREAMS=$st JLM_LSE_SHARE=$sh timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-config5 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $st share $sh: %.3f ms per step, device-resident %.3f, lse in pipeline %.1f us' % (d['ms_per_step'], d['device_resident_ms_per_step'], 1e3*d['roofline'].get('avg_launch_ms_in_pipeline',0)))"
done; done | tee gpurun_out/share_sweep.txt
