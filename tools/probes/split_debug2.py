"""debug probe: one-hot columns: relative error of single products through jlm_vocab_lse_split"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from jlm_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = None
rng = np.random.default_rng(0)
for K in (160, 256):
    k16 = (K + 15) // 16 * 16
    V, R = 1, 64
    T = (rng.standard_normal((R, K)) + 3.0).astype(np.float32)
    Tg = torch.tensor(T, device=dev)
    bad = []
    for kk in range(K):
        B = np.zeros((V, K), np.float32); B[0, kk] = np.float32(0.3 * 1.2345678)
        Bg = torch.tensor(B, device=dev)
        Bs = torch.zeros((V, k16), device=dev)
        assert L.jlm_pack_split_f16(Bg.data_ptr(), V, K, K, 64.0, Bs.data_ptr(), k16, st) == 0
        b2 = torch.zeros(V, device=dev)
        segs = (_lib.Segment * 1)(); segs[0] = _lib.Segment(0, V, K, 0, Bs.data_ptr(), k16)
        ts, ds = (ctypes.c_float * 1)(8.0), (ctypes.c_float * 1)(1.0 / 512)
        part = torch.zeros((96, R, 2), device=dev); lse = torch.zeros(R, dtype=torch.float64, device=dev)
        n = L.jlm_vocab_lse_split(segs, ts, ds, None, 1, b2.data_ptr(), Tg.data_ptr(), K, None, part.data_ptr(), R, 96, R, None, st)
        assert n > 0 and L.jlm_lse_combine(part.data_ptr(), R, n, None, lse.data_ptr(), R, None, st) == 0
        torch.cuda.synchronize()
        ref = T[:, kk].astype(np.float64) * float(B[0, kk])
        rel = np.abs(lse.cpu().numpy() - ref) / np.abs(ref)
        if rel.max() > 2e-6: bad.append((kk, float(rel.max()), int(rel.argmax())))
    print("K=%d bad columns (k, max rel err, row):" % K, bad[:40], len(bad))
