#!/bin/bash
# first GPU contact: kernel unit tests, decode parity, smoke, short bench
mkdir -p gpurun_out
python -c "import torch;print(torch.cuda.get_device_name(0))" > gpurun_out/dev.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log
tail -5 gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q --tb=short > gpurun_out/decode.log 2>&1
echo "decode rc=$?" >> gpurun_out/decode.log
tail -15 gpurun_out/decode.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log; tail -5 gpurun_out/bench.log
