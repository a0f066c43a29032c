#!/bin/bash
# round-5 end-of-milestone pass: GPU suite, smoke, rocprofv3 kernel statistics and PMC traffic of the headline workload (-> rocprof_latest.json /
# traffic_latest.json with the sources' hashes), the driver's bench command, the other configs, the parity report, the knob sweep
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
BENCH_ARGS="--no-legs --no-config5" bash tools/gpu_prof.sh prof > gpurun_out/prof_stdout.txt 2>&1; tail -14 gpurun_out/prof_stdout.txt
python tools/rocprof_report.py gpurun_out/prof/run_kernel_stats.csv gpurun_out/rocprof_latest.json > /dev/null; cp gpurun_out/rocprof_latest.json profiles/rocprof_latest.json
cp gpurun_out/prof/run_kernel_stats.csv gpurun_out/kernel_stats_configs1_only.csv
bash tools/gpu_traffic.sh; python tools/traffic_report.py gpurun_out/hbm_traffic.csv gpurun_out/traffic_latest.json | head -8; cp gpurun_out/traffic_latest.json profiles/traffic_latest.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench_driver_cmd.err; tail -1 gpurun_out/bench_driver_cmd.log | cut -c1-300
bash tools/gpu_configs.sh > gpurun_out/other_configs.log 2>&1; cat gpurun_out/other_configs.log | cut -c1-400
timeout 600 python bench.py --fixture mid-untied --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | cut -c1-400 >> gpurun_out/other_configs.log
python tools/parity_report.py > gpurun_out/parity.txt 2>&1; tail -5 gpurun_out/parity.txt
bash tools/gpu_knobs.sh > gpurun_out/knobs.txt 2>&1; cat gpurun_out/knobs.txt
