#!/bin/bash
# the decode parity tests under every off-default knob: each line must end in "passed"
for e in JLM_COLLECTOR=1 JLM_BLOCKING_SYNC=1 JLM_STREAMS=1 JLM_STREAMS=2 JLM_STREAMS=3 JLM_SIDE=0 JLM_PRECISION=f32 JLM_LSE_WAVES=4 JLM_GATE_V=0 \
         JLM_NATIVE_READOUT=0 JLM_NATIVE_LATTICE=0 JLM_LSE_MIXED=0 JLM_PINNED_LATTICE=0 JLM_GRAPH=1 JLM_LSE_SHARE=0 JLM_NO_ENV_DEFAULTS=1; do
  echo -n "$e: "; env $e python -m pytest tests/test_gpu_decode.py -x -q -k "golden or pipelined or mixed_rows" 2>&1 | tail -1
done
