#!/bin/bash
# LSTM step: the refill-in-place pipeline (JLM_GATE_V=1, default) against the round-2 loop (JLM_GATE_V=0): unit tests, kbench
# lines interleaved, per-workgroup timeline of both, ablations of the new loop (build_prof/libjlm_hip_prof.so built with
# -DJLM_PROFILE -DJLM_GATE_ABLATE)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step_xg" > gpurun_out/gate_tests.log 2>&1; tail -5 gpurun_out/gate_tests.log
for i in 1 2; do
  for v in 0 1; do
    echo "JLM_GATE_V=$v:"; JLM_GATE_V=$v timeout 600 python tools/kbench.py gate 2>&1 | grep "xg"
  done
done | tee gpurun_out/kbench_gate_ab.log
if [ -f build_prof/libjlm_hip_prof.so ]; then
  for v in 0 1; do echo "timeline JLM_GATE_V=$v"; JLM_GATE_V=$v timeout 300 python tools/probes/gate_xg_profile.py 2560; done | tee gpurun_out/gate_xg_profile_ab.log
  for abl in 1 2 4 8 6 14 5 16 22; do echo "ablation JLM_GATE_ABL=$abl (1 no MFMA, 2 no reads, 4 no DMA, 8 no barrier, 16 no epilogue loads)"; JLM_GATE_ABL=$abl timeout 300 python tools/probes/gate_xg_profile.py 2560 | grep -v "p10"; done | tee gpurun_out/gate_xg_ablation.log
fi
