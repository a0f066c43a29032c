#!/bin/bash
# round 5: the driver's bench command with the new legs, then the full GPU suite
mkdir -p gpurun_out
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_m_bench_driver_cmd.log 2> gpurun_out/r05_m_bench_driver_cmd.err
tail -1 gpurun_out/r05_m_bench_driver_cmd.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], d.get('ms_per_step_repeats'), 'dev', d.get('device_resident_ms_per_step'))
for k in ('config3', 'config4', 'config5', 'length10', 'length40', 'peaked20'):
    l = d.get(k) or {}
    print(k, l.get('value'), l.get('ms_per_step'), l.get('device_resident_ms_per_step'), (l.get('dominant_kernel') or {}).get('avg_launch_ms'), (l.get('dominant_kernel') or {}).get('lse_form'), (l.get('gate_gemm') or {}).get('mfma_util_pct'))
r = d['roofline']; print('roofline', r['frac'], r['avg_launch_ms'], r.get('split_rows_avg_launch_ms'), r.get('clock_consistency_ok'))
print('gate', d['gate_gemm']['mfma_util_pct'], d['gate_gemm']['avg_launch_ms']); print('cpu', d['cpu_baseline'])
"
tail -3 gpurun_out/r05_m_bench_driver_cmd.err
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -8 | tee gpurun_out/r05_m_gpu_tests.txt
