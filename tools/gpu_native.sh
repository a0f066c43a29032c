#!/bin/bash
# native frame loop: tests, interleaved A/B per decode kind, bench line
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for m in static static-vs dynamic; do
  f=mid-vtable; [ $m != static ] && f=mid-tied
  echo "== $f $m"; python tools/ab_engine.py $f $m 2>&1 | tail -7
done
python bench.py > gpurun_out/bench_native.json 2> gpurun_out/bench_native.err; tail -c 2500 gpurun_out/bench_native.json
