mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_edge_cases.py -m gpu -q --tb=short -x -k "untied" 2>&1 | tail -6
run() { timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(json.dumps({'ms_per_step':d['ms_per_step'],'device_resident_ms_per_step':d.get('device_resident_ms_per_step'),'lse_kernel':(r.get('kernel') or '')[:40],'lse_form':r.get('lse_form'),'calib':r.get('lse_form_calibration'),'lse_us':(r.get('avg_launch_ms') or 0)*1e3,'workload':d['config']['workload'][:50]}))"; }
for fx in mid-untied mid-tied; do echo "== $fx"; run --fixture $fx --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-legs; done | tee gpurun_out/r05_q_untied_mixed.txt
