#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_l_switch_interval.txt; : > $O
run() { timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'host_cpu_ms_per_step':d.get('host_cpu_ms_per_step'),'workload':d['config']['workload'][:60]}))"; }
for i in 1 2 3; do
for si in 0 0.0002 0.001; do
  for dec in static-vs dynamic; do
    echo "== JLM_SWITCH_INTERVAL=$si mid-tied $dec" >> $O
    JLM_SWITCH_INTERVAL=$si run --fixture mid-tied --decoder $dec --steps 40 --warmup 3 --no-cpu-baseline --no-config5 --no-legs >> $O
  done
done
done
cat $O
