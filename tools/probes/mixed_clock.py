"""Shader clock and workgroup durations of jlm_vocab_lse_mixed (a -DJLM_WGTIME build: JLM_HIP_LIB=build_prof/libjlm_hip_wgtime.so).
Usage: python tools/probes/mixed_clock.py [k ...]   (one single-segment launch per k)"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import _lib
L = _lib.lib()
WIDE = os.environ.get("JLM_MX_WIDE", "0") != "0"          # the wide kernel (jlm_mixed_w.hip) keeps its own table
reader = L.jlm_prof_read_wg_mxw if WIDE else L.jlm_prof_read_wg_mx
reader.argtypes = [ctypes.c_void_p]
ZERO = bool(os.environ.get("KBENCH_ZERO"))                 # all-zero operands: lowest switching power
dev = torch.device("cuda")
R = 2560
cases = {200: 12000, 100: 18000, 50: 20000}
ks = [int(a) for a in sys.argv[1:]] or [200, 100, 50]
for k in ks:
    V = cases.get(k, 20000)
    kp = (k + 3) // 4 * 4
    nb = (kp + 2 + 31) // 32
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
    for _ in range(int(os.environ.get("CLOCK_ITERS", "20"))):
        n = L.jlm_vocab_lse_mixed(seg, ds, s8, None, 1, Tm.data_ptr(), ld_tm, part.data_ptr(), R, 96, R, None, None)
    torch.cuda.synchronize()
    buf = np.zeros((1024, 4), dtype=np.uint64)
    assert reader(buf.ctypes.data) == 0
    w = buf[buf[:, 1] > 0].astype(np.float64)
    t0 = w[:, 0].min()
    dur = (w[:, 1] - w[:, 0]) / 100.0
    ghz = w[:, 3] / (dur * 1e3)
    print("k=%d: %d workgroups, span %.1f us; workgroup duration mean %.1f min %.1f max %.1f us; start spread %.1f us; shader clock %.2f GHz (min %.2f max %.2f)"
          % (k, len(w), (w[:, 1].max() - t0) / 100.0, dur.mean(), dur.min(), dur.max(), (w[:, 0].max() - t0) / 100.0, ghz.mean(), ghz.min(), ghz.max()))
