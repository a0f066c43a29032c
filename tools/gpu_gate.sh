#!/bin/bash
# the LSTM step: unit tests + microbenchmark of the one-tile-per-CU table form (jlm_lstm_step_xg) beside the tile form
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "lstm_step" > gpurun_out/gate_tests.log 2>&1; tail -5 gpurun_out/gate_tests.log
timeout 600 python tools/kbench.py gate 2>&1 | grep "xg\|table" | tee gpurun_out/kbench_gate.log
[ -f build_prof/libjlm_hip_prof.so ] && timeout 300 python tools/probes/gate_xg_profile.py 2560 | tee gpurun_out/gate_xg_profile.log
