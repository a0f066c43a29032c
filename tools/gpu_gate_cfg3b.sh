#!/bin/bash
# round 6: the W-stationary LSTM step's epilogue-operand schedules (in-tree WS_OPS4=1; build_prof/libjlm_hip_WSOPS{0,2,3}.so) in the REAL decode of
# BASELINE configs[2] (20 480 rows per frame) and at 10 240 rows: gate_gemm.mfma_util_pct by HIP events
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d.get('gate_gemm') or {}; print(json.dumps({'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'gate_util_pct':g.get('mfma_util_pct'),'gate_us':round((g.get('avg_launch_ms') or 0)*1e3,1)}))"; }
{
for i in 1 2 3; do
for lib in "" WSOPS0 WSOPS2 WSOPS3; do
  echo "== configs[2] JLM_GATE_V=2 ${lib:-in-tree (WS_OPS4=1)}"; JLM_GATE_V=2 JLM_HIP_LIB=${lib:+$PWD/build_prof/libjlm_hip_$lib.so} timeout 900 python bench.py --fixture big-tied --batch 1024 --beam 20 --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | line
done
done
for i in 1 2; do
for lib in "" WSOPS0 WSOPS2 WSOPS3; do
  echo "== 10 240 rows JLM_GATE_V=2 ${lib:-in-tree (WS_OPS4=1)}"; JLM_GATE_V=2 JLM_HIP_LIB=${lib:+$PWD/build_prof/libjlm_hip_$lib.so} timeout 900 python bench.py --fixture mid-tied --batch 1024 --beam 10 --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-legs 2>/dev/null | tail -1 | line
done
done
} 2>&1 | tee gpurun_out/gate_cfg3b.txt
