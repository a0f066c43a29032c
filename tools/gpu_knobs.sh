#!/bin/bash
# the decode parity tests under every off-default knob: each line must end in "passed" -- but three, by
# construction: JLM_LSE_SHARE=66 (a test asserts the default share), JLM_MIXED_MAX_LSE_RMS=0 (the load-time calibration off: the x20-peaked
# models then stay on mixed rows and miss the score tolerance -- what the calibration is for), JLM_PRECISION=f32 (plain f32 accumulation:
# 2.005e-5 against the 2e-5 tolerance on peaked20-tied/dynamic)
for e in JLM_COLLECTOR=1 JLM_BLOCKING_SYNC=1 JLM_STREAMS=1 JLM_STREAMS=2 JLM_STREAMS=3 JLM_SIDE=0 JLM_PRECISION=f32 JLM_LSE_WAVES=4 JLM_GATE_V=0 JLM_GATE_V=2 JLM_GATE_V=3 JLM_LSE_SHARE=66 JLM_MIXED_MAX_LSE_RMS=0 JLM_SUBMIT_THREAD=1 JLM_NUMA_PIN=0 \
         JLM_NATIVE_READOUT=0 JLM_NATIVE_LATTICE=0 JLM_LSE_MIXED=0 JLM_PINNED_LATTICE=0 JLM_GRAPH=1 JLM_LSE_SHARE=0 JLM_NO_ENV_DEFAULTS=1; do
  echo -n "$e: "; env $e python -m pytest tests/test_gpu_decode.py -x -q -k "golden or pipelined or mixed_rows" 2>&1 | tail -1
done
