mkdir -p gpurun_out
JLM_PRECISION=f32 python -m pytest tests/test_gpu_decode.py tests/test_gpu_kernels.py -x -q --tb=short -k "golden or pipelined or mixed_rows or identical_rows" > gpurun_out/dbg_f32.log 2>&1; tail -40 gpurun_out/dbg_f32.log
timeout 1800 python -m pytest tests/test_gpu_kernels.py -q --tb=short -m gpu -k "beam_step" > gpurun_out/dbg_beam.log 2>&1; tail -30 gpurun_out/dbg_beam.log
