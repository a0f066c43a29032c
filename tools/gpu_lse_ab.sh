mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "vocab_lse_split or lse" > gpurun_out/lse_tests.log 2>&1; tail -3 gpurun_out/lse_tests.log
KBENCH_ONLY=split bash tools/ab_libs.sh lse 3 "vocab_lse_split" 2>&1 | tee gpurun_out/lse_ab.log
