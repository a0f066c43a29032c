#!/bin/bash
# round 5: strings -> strings vs device-resident rates (bench.py) of the selected-vocabulary decoders and the headline, enqueue on the
# calling thread (JLM_SUBMIT_THREAD=0) or on its own thread (=1)
mkdir -p gpurun_out
O=gpurun_out/r05_k_host_bench.txt; : > $O
run() { timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'host_cpu_ms_per_step':d.get('host_cpu_ms_per_step'),'workload':d['config']['workload'][:60]}))"; }
for i in 1 2; do
for st in 0 1; do
  for dec in static-vs dynamic; do
    echo "== JLM_SUBMIT_THREAD=$st mid-tied $dec" >> $O
    JLM_SUBMIT_THREAD=$st run --fixture mid-tied --decoder $dec --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-legs >> $O
  done
  echo "== JLM_SUBMIT_THREAD=$st headline" >> $O
  JLM_SUBMIT_THREAD=$st run --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-legs >> $O
done
done
cat $O
