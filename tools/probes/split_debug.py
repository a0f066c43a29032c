"""debug probe: per-k-step error of jlm_vocab_lse_split (V=1 row -> lse = that row's logit)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from jlm_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = None
rng = np.random.default_rng(0)
for K in (64, 112, 160, 200, 256):
    k16 = (K + 15) // 16 * 16
    V, R = 300, 40
    res = []
    for step in list(range(k16 // 16)) + [-1]:
        B = np.zeros((V, K), np.float32)
        if step < 0: B[:] = rng.standard_normal((V, K)) * 0.3
        else: B[:, 16*step:16*step+16] = rng.standard_normal((V, min(16, K-16*step))) * 0.3
        T = rng.standard_normal((R, K)).astype(np.float32)
        Bg, Tg = torch.tensor(B, device=dev), torch.tensor(T, device=dev)
        Bs = torch.zeros((V, k16), device=dev)
        assert L.jlm_pack_split_f16(Bg.data_ptr(), V, K, K, 64.0, Bs.data_ptr(), k16, st) == 0
        b2 = torch.zeros(V, device=dev)
        segs = (_lib.Segment * 1)(); segs[0] = _lib.Segment(0, V, K, 0, Bs.data_ptr(), k16)
        ts, ds = (ctypes.c_float * 1)(8.0), (ctypes.c_float * 1)(1.0 / 512)
        part = torch.zeros((96, R, 2), device=dev); lse = torch.zeros(R, dtype=torch.float64, device=dev)
        n = L.jlm_vocab_lse_split(segs, ts, ds, None, 1, b2.data_ptr(), Tg.data_ptr(), K, None, part.data_ptr(), R, 96, R, None, st)
        assert n > 0 and L.jlm_lse_combine(part.data_ptr(), R, n, None, lse.data_ptr(), R, None, st) == 0
        torch.cuda.synchronize()
        y = T.astype(np.float64) @ B.astype(np.float64).T
        mx = y.max(1); ref = mx + np.log(np.exp(y - mx[:, None]).sum(1))
        res.append("%s:%.1e" % (step, np.abs(lse.cpu().numpy() - ref).max()))
    print("K=%d parts=%d  " % (K, n) + " ".join(res))
