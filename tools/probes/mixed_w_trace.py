"""Round 5: where a workgroup of the WIDE mixed vocabulary kernel spends its cycles (a -DMXW_TRACE build: JLM_HIP_LIB=build_prof/
libjlm_hip_TR.so, JLM_MX_WIDE=1).  One single-segment launch per k; per workgroup: prologue (row operands + first tile), every
tile, epilogue, in shader-clock cycles, against the cycles the tile's matrix instructions need (32 each).
Usage: python tools/probes/mixed_w_trace.py [k ...]"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import _lib
L = _lib.lib()
L.jlm_prof_read_mxw_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
R = 2560
ZERO = bool(os.environ.get("KBENCH_ZERO"))
cases = {200: 12000, 100: 18000, 50: 20000}
for k in [int(a) for a in sys.argv[1:]] or [200, 100, 50]:
    V = cases[k]
    kp = (k + 3) // 4 * 4
    nb = (kp + 2 + 31) // 32
    ns16 = (kp + 2 + 15) // 16
    mtt = 2 if nb >= 5 else 4 if nb >= 3 else 8
    B = torch.randn(V, kp, device=dev) * (0.0 if ZERO else 0.05)
    b2 = torch.randn(V, device=dev) * 0.05
    dst = torch.zeros((V, 32 * nb), device=dev)
    assert L.jlm_pack_mixed(B.data_ptr(), V, kp, kp, b2.data_ptr(), 2.0 ** 15, 2.0 ** 15 * 1.4427, 2.0 ** 7, dst.data_ptr(), 32 * nb, None) == 0
    seg = (_lib.Segment * 1)(_lib.Segment(0, V, kp, 0, dst.data_ptr(), 32 * nb))
    ts, ds, s8 = (ctypes.c_float * 1)(2.0 ** 10), (ctypes.c_float * 1)(2.0 ** -25), (ctypes.c_float * 1)(2.0 ** 7)
    T = torch.randn(R, kp, device=dev) * (0.0 if ZERO else 0.3)
    ld_tm = L.jlm_mixed_t_stride(seg, 1)
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), device=dev)
    assert L.jlm_pack_t_mixed(seg, ts, 1, T.data_ptr(), kp, None, R, None, Tm.data_ptr(), ld_tm, None) == 0
    part = torch.empty((96, R, 2), device=dev)
    for _ in range(20):
        n = L.jlm_vocab_lse_mixed(seg, ds, s8, None, 1, Tm.data_ptr(), ld_tm, part.data_ptr(), R, 96, R, None, None)
    torch.cuda.synchronize()
    buf = np.zeros((256, 64), dtype=np.uint64)
    assert L.jlm_prof_read_mxw_trace(buf.ctypes.data) == 0
    w = buf[:240].astype(np.float64)
    pro = w[:, 1] - w[:, 0]
    ntile = ((w[:, 2:62] > 0).sum(axis=1)).astype(int)
    tiles = []
    for b in range(240):
        st = np.concatenate([[w[b, 1]], w[b, 2:2 + ntile[b]]])
        tiles.append(np.diff(st))
    full = np.concatenate([t[:-1] for t in tiles if len(t) > 1])
    last = np.array([t[-1] for t in tiles])
    epi = np.array([w[b, 63] - w[b, 1 + ntile[b]] for b in range(240)])
    total = w[:, 63] - w[:, 0]
    mf = 2 * (ns16 + 2 * nb) * mtt * 32
    print("k=%d: tiles per workgroup %d..%d; prologue %.0f cycles (min %.0f max %.0f); a whole tile %.0f cycles (min %.0f max %.0f) against %d for its %d matrix "
          "instructions = %.0f %%; last tile %.0f; epilogue %.0f; workgroup total %.0f (max %.0f)"
          % (k, ntile.min(), ntile.max(), pro.mean(), pro.min(), pro.max(), full.mean(), full.min(), full.max(), mf, mf // 32, 100.0 * mf / full.mean(),
             last.mean(), epi.mean(), total.mean(), total.max()))
