#!/bin/bash
# one-off reruns (round 6: the wave-per-sentence backtrace): beam-step / backtrace kernel tests, the decode goldens, its launch time against the thread-per-path kernel
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode.py tests/test_gpu_edge_cases.py -q --tb=short -m gpu -k "beam_step or golden or pipelined or traces or oversized or long" > gpurun_out/dbg_backtrace.log 2>&1; tail -5 gpurun_out/dbg_backtrace.log
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/bt$w
  JLM_BACKTRACE_WAVE=$w timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bt$w -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-legs --no-config5 > /dev/null 2>&1
  echo "JLM_BACKTRACE_WAVE=$w:"; grep -h "backtrace" $GRAFT_REPO_ROOT/gpurun_out/bt$w/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/bt$w/*kernel_stats.csv 2>/dev/null | cut -c1-200
done
