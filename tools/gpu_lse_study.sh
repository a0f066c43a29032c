#!/bin/bash
# LSE kernel study for profiles/: per-workgroup timeline + shader clock at 24 / 16 / 8 columns (needs build_prof/libjlm_hip_wgtime.so:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_WGTIME -Iinclude -Ijlm_amd/csrc -o build_prof/libjlm_hip_wgtime.so \
#     jlm_amd/csrc/jlm_{gemm,beam,split,gate,decode}.hip ), the pure-MFMA clock probe, kbench lines.
mkdir -p gpurun_out
{
for n in 24 16 8; do echo "== JLM_LSE_NP=$n (columns; x 10 row tiles = workgroups)"; JLM_PROF_LIB=libjlm_hip_wgtime.so JLM_LSE_NP=$n timeout 120 python tools/probes/lse_wg_timeline.py 2>&1 | grep -E "parts|seg |shader clock"; done
} > gpurun_out/lse_wg_timeline.txt 2>&1
cat gpurun_out/lse_wg_timeline.txt
hipcc --offload-arch=gfx950 -O2 -o /tmp/mcc tools/probes/mfma_clock_vs_cus.hip 2>/dev/null && timeout 120 /tmp/mcc > gpurun_out/mfma_clock_vs_cus.txt 2>&1; head -9 gpurun_out/mfma_clock_vs_cus.txt
timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_s" > gpurun_out/stat_lse.log; cat gpurun_out/stat_lse.log
