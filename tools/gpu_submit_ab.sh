#!/bin/bash
# strings -> strings rate with the enqueue on its own thread (JLM_SUBMIT_THREAD=1, round 4) and on the calling thread (0), per decoder kind
mkdir -p gpurun_out
for dec in static static-vs dynamic; do
  fx=mid-vtable; [ $dec = dynamic ] && fx=mid-tied
  for st in 0 1 0 1; do
    JLM_SUBMIT_THREAD=$st timeout 300 python bench.py --steps 20 --warmup 5 --decoder $dec --fixture $fx --no-legs --no-config5 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dec submit_thread=$st: %.3f ms per step strings->strings, %.3f device-resident, host cpu %.2f ms per step' % (d['ms_per_step'], d['device_resident_ms_per_step'], d['host_cpu_ms_per_step']))"
  done
done | tee gpurun_out/submit_ab.txt
