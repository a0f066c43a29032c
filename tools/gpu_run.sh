#!/bin/bash
# One parametrised entry for the GPU box (replaces the per-round gpu_roundN.sh / gpu_r05_*.sh scripts; their steps are the words below).
#   gpurun -- 'bash tools/gpu_run.sh tests smoke prof traffic bench configs parity knobs'
# Steps write under gpurun_out/ (scratch); summaries worth keeping are copied into profiles/ by hand or by the step itself where it says so.
#   tests    the whole GPU suite                          smoke    __graft_entry__.smoke()
#   prof     rocprofv3 --kernel-trace --stats of the headline bench -> gpurun_out/prof, profiles/rocprof_latest.json (hash-tagged)
#   traffic  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/traffic_latest.json (hash-tagged)
#   bench    the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5)
#   configs  the other BASELINE configs + the untied model, one abridged line each (tools/gpu_configs.sh)
#   parity   every golden case through the HIP path (tools/parity_report.py)
#   knobs    the decode parity tests under every off-default environment knob (tools/gpu_knobs.sh)
#   kbench   tools/kbench.py (all kernels)            mx6ab   the two cross-term formats side by side, interleaved (KBENCH_MX6)
#   ablate   the mx6 kernel against build_prof/libjlm_hip_<TAG>.so ablation builds (tools/gpu_r06_abl.sh)
#   gatepmc  SQ / TCC counters of the LSTM step (tools/gpu_gate_pmc.sh)   lsepmc  ... of the vocabulary kernel (tools/gpu_pmc.sh)
mkdir -p gpurun_out
for step in "$@"; do
  echo "=== $step"
  case $step in
    tests)   timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log ;;
    smoke)   timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log ;;
    prof)    BENCH_ARGS="--no-legs --no-config5" bash tools/gpu_prof.sh prof > gpurun_out/prof_stdout.txt 2>&1; tail -14 gpurun_out/prof_stdout.txt
             python tools/rocprof_report.py gpurun_out/prof/run_kernel_stats.csv gpurun_out/rocprof_latest.json > /dev/null && cp gpurun_out/rocprof_latest.json profiles/rocprof_latest.json
             cp gpurun_out/prof/run_kernel_stats.csv gpurun_out/kernel_stats_configs1_only.csv ;;
    traffic) bash tools/gpu_traffic.sh; python tools/traffic_report.py gpurun_out/hbm_traffic.csv gpurun_out/traffic_latest.json | head -8; cp gpurun_out/traffic_latest.json profiles/traffic_latest.json ;;
    bench)   timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench_driver_cmd.err; tail -1 gpurun_out/bench_driver_cmd.log | cut -c1-300 ;;
    configs) bash tools/gpu_configs.sh > gpurun_out/other_configs.log 2>&1; cut -c1-400 gpurun_out/other_configs.log
             timeout 600 python bench.py --fixture mid-untied --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | cut -c1-400 >> gpurun_out/other_configs.log ;;
    parity)  python tools/parity_report.py > gpurun_out/parity.txt 2>&1; tail -5 gpurun_out/parity.txt ;;
    knobs)   bash tools/gpu_knobs.sh > gpurun_out/knobs.txt 2>&1; cat gpurun_out/knobs.txt ;;
    kbench)  timeout 900 python tools/kbench.py > gpurun_out/kbench.txt 2>&1; grep -v "^parts" gpurun_out/kbench.txt | cut -c1-140 ;;
    mx6ab)   KBENCH_MX6=2 timeout 600 python tools/kbench.py > gpurun_out/kbench_mx6.txt 2>&1; grep -v "^parts" gpurun_out/kbench_mx6.txt | cut -c1-140 ;;
    ablate)  bash tools/gpu_r06_abl.sh > gpurun_out/mx6_ablate.txt 2>&1; cut -c1-120 gpurun_out/mx6_ablate.txt ;;
    gatepmc) bash tools/gpu_gate_pmc.sh ;;
    lsepmc)  bash tools/gpu_pmc.sh ;;
    *) echo "unknown step $step" ;;
  esac
done
