#!/bin/bash
# round 5: (1) the driver's bench command; (2) untied vs tied step time; (3) eight ranks with real kernels sharing ONE GPU (the host
# side of BASELINE configs[4] under the box's 16-CPU quota: 2 CPUs per rank)
mkdir -p gpurun_out
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_n_bench_driver_cmd.log 2> gpurun_out/r05_n_bench_driver_cmd.err
tail -1 gpurun_out/r05_n_bench_driver_cmd.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], d.get('ms_per_step_repeats'), 'dev', d.get('device_resident_ms_per_step'))
for k in ('config3', 'config4', 'config5', 'length10', 'length40', 'peaked20'):
    l = d.get(k) or {}
    print(k, l.get('value'), l.get('ms_per_step'), l.get('device_resident_ms_per_step'), l.get('error'))
"
tail -4 gpurun_out/r05_n_bench_driver_cmd.err
run() { timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(json.dumps({'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'lse_kernel':(r.get('kernel') or '')[:60],'lse_us':(r.get('avg_launch_ms') or 0)*1e3,'workload':d['config']['workload'][:60]}))"; }
for fx in mid-untied mid-tied; do echo "== $fx"; run --fixture $fx --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-legs; done | tee gpurun_out/r05_n_untied.txt
echo "== 8 ranks on one GPU" | tee gpurun_out/r05_n_shared8.txt
for n in 1 8; do
  timeout 900 python bench.py --gpus $n --debug-shared-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-config5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'n_ranks': d['n_gpus'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'repeats': d.get('ms_per_step_repeats'), 'host_cpu_ms_per_step_rank0': d.get('host_cpu_ms_per_step'), 'host_cpus': d.get('host_cpus')}))
" | tee -a gpurun_out/r05_n_shared8.txt
done
