#!/bin/bash
# round 6: the LSTM step's forms in the REAL decode of BASELINE configs[2] (1 024 sentences x beam 20, V = 100 k: 20 480 rows per frame) and of
# configs[4]'s per-GPU share shape (10 240 rows): gate_gemm.mfma_util_pct by HIP events, per JLM_GATE_V
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d.get('gate_gemm') or {}; print(json.dumps({'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'gate_util_pct':g.get('mfma_util_pct'),'gate_us':round((g.get('avg_launch_ms') or 0)*1e3,1),'rows':g.get('rows_per_launch')}))"; }
{
for i in 1 2; do
for v in "" 2 3 4; do
  echo "== configs[2] JLM_GATE_V=${v:-default}"; JLM_GATE_V=$v timeout 900 python bench.py --fixture big-tied --batch 1024 --beam 20 --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | line
done
done
for v in "" 2 3 4; do
  echo "== 1 024 sentences x beam 10 (10 240 rows) JLM_GATE_V=${v:-default}"; JLM_GATE_V=$v timeout 900 python bench.py --fixture mid-tied --batch 1024 --beam 10 --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | line
done
} 2>&1 | tee gpurun_out/gate_cfg3.txt
