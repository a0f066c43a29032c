#!/bin/bash
# round 5: the D-softmax* launch without a running maximum (eight-wave kernel, mx_body FR): tests, then kbench and the headline A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -x --tb=short -k "identical_rows or test_vocab_lse_mixed" 2>&1 | tail -3
python -m pytest tests/test_gpu_decode.py tests/test_gpu_mixed_logits.py -q -x --tb=short 2>&1 | tail -3
run() { timeout 900 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(json.dumps({'ms_per_step':d['ms_per_step'],'repeats':d.get('ms_per_step_repeats'),'dev':d.get('device_resident_ms_per_step'),'lse_us':(r.get('avg_launch_ms') or 0)*1e3,'fixed_ref':(r.get('lse_form_calibration') or {}).get('fixed_ref')}))"; }
for i in 1 2 3; do for f in 1 0; do echo "JLM_MX_FIXREF=$f"; JLM_MX_FIXREF=$f run --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-legs; done; done | tee gpurun_out/r05_v_fixed_ref_dsoftmax.txt
