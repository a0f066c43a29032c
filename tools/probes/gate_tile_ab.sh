# LSTM-step tile forms: correctness of each (kernel tests), then interleaved timing
for t in 160 128; do echo "== tests tile=$t"; JLM_GATE_TILE=$t timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lstm_step" 2>&1 | tail -2; done
for rep in 1 2 3; do for t in 128 160; do export JLM_GATE_TILE=$t; echo -n "tile=$t  "; timeout 300 python tools/kbench.py gate 2>&1 | grep "xgate-table H=512 E=200"; done; done
