#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "wordlist" > gpurun_out/kernels.log 2>&1; tail -12 gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_edge_cases.py -m gpu -q --tb=short > gpurun_out/decode.log 2>&1; tail -6 gpurun_out/decode.log
run() { echo "== $*"; timeout 900 python bench.py "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'e2e':d['end_to_end_chars_per_s']}))"; }
run --fixture mid-tied --decoder dynamic --steps 5 --warmup 2 --no-cpu-baseline
run --fixture mid-tied --decoder static-vs --steps 5 --warmup 2 --no-cpu-baseline
