#!/bin/bash
# round 5: a segment's head / the first segment on split rows beside mixed rows (jlm_vocab_lse_hybrid, ABI 10): kernel tests, the loader's
# gates, the peaked20 decodes, the launch forms timed side by side, the peaked20 leg
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "hybrid or lse_mixed" --tb=short 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_mixed_logits.py -q -x --tb=short 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_decode.py -q -x -k "vtable or wideh" --tb=short 2>&1 | tail -5
KBENCH_ONLY=split timeout 600 python tools/kbench.py lse 2>&1 | grep -v "^parts" | tee gpurun_out/head_kbench.txt
for fx in peaked20-vtable mid-vtable; do
  timeout 600 python bench.py --fixture $fx --steps 20 --warmup 3 --no-legs --no-config5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/head_bench_$fx.json
  python - <<P
import json
d = json.load(open("gpurun_out/head_bench_$fx.json"))
r = d["roofline"]
print("$fx", d["ms_per_step"], d.get("device_resident_ms_per_step"), "lse_us", r.get("avg_launch_ms"), r.get("kernel", "")[:60], r.get("lse_form"), r.get("lse_form_calibration"))
P
done
