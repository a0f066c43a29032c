#!/bin/bash
# the character-model decoder on the GPU: its tests, then a throughput line (tools/probes/char_bench.py)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_char.py -x -q -m gpu 2>&1 | tail -30
[ -f tools/probes/char_bench.py ] && timeout 600 python tools/probes/char_bench.py 2>&1 | tail -20
