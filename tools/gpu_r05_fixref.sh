mkdir -p gpurun_out

python -m pytest tests/test_gpu_decode.py -q -x --tb=short -k "tied or untied" 2>&1 | tail -3
run() { timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(json.dumps({'ms_per_step':d['ms_per_step'],'dev':d.get('device_resident_ms_per_step'),'lse_us':(r.get('avg_launch_ms') or 0)*1e3,'calib':r.get('lse_form_calibration'),'w':d['config']['workload'][:40]}))"; }
for i in 1 2; do for f in 1 0; do echo "JLM_MX_FIXREF=$f"; JLM_MX_FIXREF=$f run --fixture mid-tied --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-legs; JLM_MX_FIXREF=$f run --fixture big-tied --batch 1024 --beam 20 --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-legs; done; done | tee gpurun_out/r05_u_fixed_ref.txt
