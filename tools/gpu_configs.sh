#!/bin/bash
# the other BASELINE.json configs (numbers for DESIGN.md), one JSON line each
mkdir -p gpurun_out
run() { echo "== $*"; timeout 900 python bench.py "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; g=d.get('gate_gemm') or {}; c=d.get('cpu_baseline') or {}; print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'e2e':d['end_to_end_chars_per_s'],'lse_tflops':r.get('achieved'),'lse_frac':r.get('frac'),'gate_tflops':g.get('achieved'),'cpu':c.get('value'),'cpu_sample':(c.get('sample') or '')[-60:]}))"; }
run --fixture mid-tied --steps 20 --warmup 3 --cpu-sentences 8 > gpurun_out/cfg_tied.log 2>&1; cat gpurun_out/cfg_tied.log
run --fixture big-tied --batch 1024 --beam 20 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/cfg3.log 2>&1; cat gpurun_out/cfg3.log
run --fixture mid-tied --decoder dynamic --steps 20 --warmup 3 --cpu-sentences 16 > gpurun_out/cfg4.log 2>&1; cat gpurun_out/cfg4.log
run --fixture mid-tied --decoder static-vs --steps 20 --warmup 3 --cpu-sentences 16 > gpurun_out/cfg_vs.log 2>&1; cat gpurun_out/cfg_vs.log
run --fixture mid-tied --batch 1024 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/cfg5.log 2>&1; cat gpurun_out/cfg5.log
