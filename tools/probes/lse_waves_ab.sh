for w in 8 4; do export JLM_LSE_WAVES=$w; echo "waves=$w"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "split" 2>&1 | tail -1; done
for rep in 1 2; do for w in 8 4; do echo "waves=$w"; JLM_LSE_WAVES=$w timeout 300 python tools/kbench.py lse 2>&1 | grep "vocab_lse_split"; done; done
