#!/bin/bash
# tied k = 256 instantiation without spills: kernel tests, golden decodes of the tied fixtures, interleaved A/B against the
# previous build (build_prof/libjlm_hip_b.so), then host profiles of the selected-vocabulary decoders
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mixed_logits.py -m gpu -q --tb=short -k "mixed or lse" > gpurun_out/tied_tests.log 2>&1; tail -3 gpurun_out/tied_tests.log
timeout 1500 python -m pytest tests/test_gpu_decode.py -m gpu -q --tb=short -k "tied" > gpurun_out/tied_decode.log 2>&1; tail -3 gpurun_out/tied_decode.log
KBENCH_ONLY=tied50k bash tools/ab_lib.sh lse 3 "vocab_lse_mixed" 2>&1 | tee gpurun_out/tied_ab.log
for m in static-vs dynamic; do
  timeout 600 python tools/probes/e2e_profile.py mid-tied $m 40 > gpurun_out/e2e_$m.log 2>&1; head -60 gpurun_out/e2e_$m.log
done
