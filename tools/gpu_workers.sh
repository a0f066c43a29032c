#!/bin/bash
# round 6: lattice / vocabulary-list worker threads (JLM_PREFETCH_WORKERS) against the strings -> strings rate of the vocab_select decoder
# (interleaved repeats), the incremental decoder (BASELINE configs[3]) and the headline
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'host_cpus':(d.get('host_cpus') or {}).get('lattice_workers')}))"; }
{
for i in 1 2 3; do
for w in 3 4 5; do
  echo "== static-vs, JLM_PREFETCH_WORKERS=$w"; JLM_PREFETCH_WORKERS=$w timeout 600 python bench.py --fixture mid-tied --decoder static-vs --steps 40 --warmup 3 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | line
done
done
for w in 3 5; do
  echo "== dynamic, JLM_PREFETCH_WORKERS=$w"; JLM_PREFETCH_WORKERS=$w timeout 600 python bench.py --fixture mid-tied --decoder dynamic --steps 40 --warmup 3 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | line
done
} 2>&1 | tee gpurun_out/workers2.txt
