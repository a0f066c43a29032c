#!/bin/bash
# round-4 end-of-milestone pass: GPU suite, smoke, the driver's bench command, the default bench, rocprofv3 kernel statistics and PMC traffic of
# the headline workload (-> rocprof_latest.json / traffic_latest.json with the sources' hashes), the other configs, the parity report
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
BENCH_ARGS="--no-legs --no-config5" bash tools/gpu_prof.sh prof > gpurun_out/prof_stdout.txt 2>&1; tail -12 gpurun_out/prof_stdout.txt
python tools/rocprof_report.py gpurun_out/prof/run_kernel_stats.csv gpurun_out/rocprof_latest.json > /dev/null; cp gpurun_out/rocprof_latest.json profiles/rocprof_latest.json
bash tools/gpu_traffic.sh; python tools/traffic_report.py gpurun_out/hbm_traffic.csv gpurun_out/traffic_latest.json | head -8; cp gpurun_out/traffic_latest.json profiles/traffic_latest.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench_driver_cmd.err; tail -1 gpurun_out/bench_driver_cmd.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-300
bash tools/gpu_configs.sh > gpurun_out/other_configs.log 2>&1; cat gpurun_out/other_configs.log | cut -c1-400
python tools/parity_report.py > gpurun_out/parity.txt 2>&1; tail -5 gpurun_out/parity.txt
