#!/bin/bash
# the other BASELINE.json configs (numbers for DESIGN.md), one abridged JSON line each
mkdir -p gpurun_out
run() { echo "== bench.py $*"; timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; g=d.get('gate_gemm') or {}; c=d.get('cpu_baseline') or {}; print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'device_resident_chars_per_s':d.get('device_resident_chars_per_s'),'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'workload':d['config']['workload'][:70],'lse_tflops':r.get('achieved'),'lse_frac':r.get('frac'),'lse_us':(r.get('avg_launch_ms') or 0)*1e3,'gate_util_pct':g.get('mfma_util_pct'),'gate_us':(g.get('avg_launch_ms') or 0)*1e3,'cpu':c.get('value')}))"; }
run --fixture mid-tied --steps 20 --warmup 3 --cpu-sentences 8 --no-config5 > gpurun_out/cfg_tied.log 2>&1; cat gpurun_out/cfg_tied.log
run --fixture big-tied --batch 1024 --beam 20 --steps 5 --warmup 2 --no-cpu-baseline --no-config5 > gpurun_out/cfg3.log 2>&1; cat gpurun_out/cfg3.log
run --fixture mid-tied --decoder dynamic --steps 20 --warmup 3 --cpu-sentences 16 --no-config5 > gpurun_out/cfg4.log 2>&1; cat gpurun_out/cfg4.log
run --fixture mid-tied --decoder static-vs --steps 20 --warmup 3 --cpu-sentences 16 --no-config5 > gpurun_out/cfg_vs.log 2>&1; cat gpurun_out/cfg_vs.log
run --config 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/cfg5.log 2>&1; cat gpurun_out/cfg5.log
