#!/bin/bash
mkdir -p gpurun_out
for cfg in "1 1" "1 0" "0 1" "0 0"; do set -- $cfg; echo "GRAPH=$1 SIDE=$2"; JLM_GRAPH=$1 JLM_SIDE=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_eager_with_events'], d['end_to_end_chars_per_s'])"; done
