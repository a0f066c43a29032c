for rep in 1 2; do for t in 64 128 256; do export JLM_GATE_TILE=$t; echo "tile=$t"; timeout 300 python tools/kbench.py gate 2>&1 | grep "xgate-table H=512 E=200"; done; done
