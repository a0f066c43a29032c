#!/bin/bash
# round 5: one-op enqueue (decode_batch): decode parity tests, then host profiles and strings -> strings rates of the three decoder
# kinds with the enqueue on the calling thread (JLM_SUBMIT_THREAD=0) and on its own thread (=1), interleaved
mkdir -p gpurun_out
O=gpurun_out/r05_j_host.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_edge_cases.py tests/test_gpu_shard.py -m gpu -q --tb=short -x 2>&1 | tail -5 >> $O
for i in 1 2; do
for k in static-vs dynamic static; do
  for st in 0 1; do
    echo "== $k JLM_SUBMIT_THREAD=$st" >> $O
    JLM_SUBMIT_THREAD=$st timeout 300 python tools/probes/host_profile2.py $k 2>&1 | grep -A1 "round [12]" >> $O
  done
done
done
cat $O
