#!/bin/bash
# round 6: the LSTM-step kernels with the pre-activation's scale folded into the exponent's constant (in-tree) against the build before
# (build_prof/libjlm_hip_OLD.so): bit-identical results expected (the scale is a power of two), one multiply less per gate
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "lstm_step" > gpurun_out/gate_fold_tests.log 2>&1; tail -3 gpurun_out/gate_fold_tests.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jlm_amd import _lib
new = _lib.lib()
old = ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "build_prof", "libjlm_hip_OLD.so"))
for name in ("jlm_lstm_step_xg", "jlm_pack_split_f16"):
    getattr(old, name).argtypes = getattr(new, name).argtypes; getattr(old, name).restype = getattr(new, name).restype
dev = torch.device("cuda:0"); H, V = 512, 3000
g = torch.Generator(device=dev); g.manual_seed(5)
rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev, generator=g) * scale
for R in (700, 2560, 5200, 10240, 20480):
    G = 3 * R
    hf, c0 = torch.tanh(rnd(G, H)), rnd(G, H)
    wtf = rnd(4 * H, H, scale=0.05)
    h0, wt = torch.zeros_like(hf), torch.zeros_like(wtf)
    assert new.jlm_pack_split_f16(hf.data_ptr(), G, H, H, 2.0 ** 14, h0.data_ptr(), H, None) == 0
    assert new.jlm_pack_split_f16(wtf.data_ptr(), 4 * H, H, H, 2.0 ** 6, wt.data_ptr(), H, None) == 0
    xg = rnd(V, 4 * H, scale=2.0 ** 20)
    rows = (torch.arange(R, device=dev, dtype=torch.int32) + 2 * R).contiguous()
    prev = torch.randint(-1, 2 * R, (G,), device=dev, dtype=torch.int32, generator=g)
    word = torch.randint(0, V, (G,), device=dev, dtype=torch.int32, generator=g)
    nd = torch.tensor([R - 3], device=dev, dtype=torch.int32)
    outs = []
    for L in (new, old):
        h, c = h0.clone(), c0.clone()
        assert L.jlm_lstm_step_xg(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows.data_ptr(), prev.data_ptr(), word.data_ptr(),
                                  wt.data_ptr(), xg.data_ptr(), H, 2.0 ** -20, 2.0 ** 14, None, R, nd.data_ptr(), None) == 0
        torch.cuda.synchronize(); outs.append((h.view(torch.int32).cpu().numpy(), c.view(torch.int32).cpu().numpy()))
    print("R = %5d: h bit-identical %s, c bit-identical %s" % (R, np.array_equal(outs[0][0], outs[1][0]), np.array_equal(outs[0][1], outs[1][1])))
PY
{
for i in 1 2 3; do
  echo "== in-tree (folded)"; KBENCH_WORDS=zipf timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
  echo "== OLD"; KBENCH_WORDS=zipf JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_OLD.so timeout 300 python tools/kbench.py gate 2>&1 | grep "lstm_step_xg"
done
} 2>&1 | tee gpurun_out/gate_fold_kbench.txt
