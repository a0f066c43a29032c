#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q --tb=short -x > gpurun_out/decode.log 2>&1; tail -3 gpurun_out/decode.log
run() { echo "== $*"; timeout 900 python bench.py "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; g=d.get('gate_gemm') or {}; c=d.get('cpu_baseline') or {}; print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'e2e':d['end_to_end_chars_per_s'],'lse_tflops':r.get('achieved'),'gate_tflops':g.get('achieved'),'cpu':c.get('value')}))"; }
run --fixture mid-tied --decoder dynamic --steps 5 --warmup 2 --no-cpu-baseline
run --fixture mid-tied --decoder static-vs --steps 5 --warmup 2 --no-cpu-baseline
