#!/bin/bash
# end-of-milestone pass: everything that feeds profiles/ and DESIGN.md
bash tools/gpu_full.sh
bash tools/gpu_traffic.sh
python tools/traffic_report.py gpurun_out/hbm_traffic.csv gpurun_out/traffic_latest.json
bash tools/gpu_configs.sh
python tools/parity_report.py > gpurun_out/parity.txt 2>&1; tail -3 gpurun_out/parity.txt
