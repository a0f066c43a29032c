#!/bin/bash
# the decode parity tests under every off-default knob that is left: each line must end in "passed"
# (round 5: the tests know the knobs -- a case a knob is known to move out of the default bars is skipped or relaxed there, not here)
for e in JLM_STREAMS=1 JLM_STREAMS=2 JLM_SIDE=0 JLM_PRECISION=f32 JLM_LSE_WAVES=4 JLM_GATE_V=2 JLM_GATE_V=3 JLM_GATE_V=4 JLM_BEAM_CHUNK=96 JLM_BACKTRACE_WAVE=0 JLM_MX_WIDE=1 JLM_NUMA_PIN=0 \
         JLM_NATIVE_READOUT=0 JLM_NATIVE_LATTICE=0 JLM_LSE_MIXED=0 JLM_LSE_MX6=0 JLM_MX6_WIDE=1 JLM_MX6_WIDE=0 JLM_MX_FIXREF=0; do
  echo -n "$e: "; env $e python -m pytest tests/test_gpu_decode.py tests/test_gpu_kernels.py -x -q -k "golden or pipelined or mixed_rows or identical_rows" 2>&1 | tail -1
done
