#!/bin/bash
for side in 1 0 1 0; do echo "SIDE=$side"; JLM_SIDE=$side timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['end_to_end_chars_per_s'])"; done
