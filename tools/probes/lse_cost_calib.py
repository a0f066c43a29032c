"""calibrate the per-vocabulary-row cost of the split LSE kernel as a function of the k-steps ns
(single-segment launches, R = 2560, two vocabulary sizes each -> slope = cost per row)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from jlm_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
def t_us(V, K, iters=30):
    k16 = (K + 15) // 16 * 16
    Bm = torch.randn(V, K, device=dev) * 0.05
    Bs = torch.zeros((V, k16), device=dev)
    assert L.jlm_pack_split_f16(Bm.data_ptr(), V, K, K, 1024.0, Bs.data_ptr(), k16, None) == 0
    segs = (_lib.Segment * 1)(); segs[0] = _lib.Segment(0, V, K, 0, Bs.data_ptr(), k16)
    ts, ds = (ctypes.c_float * 1)(16.0), (ctypes.c_float * 1)(1.0 / 16384)
    T, b2 = torch.randn(R, K, device=dev), torch.randn(V, device=dev) * 0.05
    part = torch.empty((96, R, 2), device=dev)
    f = lambda: L.jlm_vocab_lse_split(segs, ts, ds, None, 1, b2.data_ptr(), T.data_ptr(), K, None, part.data_ptr(), R, 96, R, None, None)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
for K in (32, 48, 64, 112, 160, 208, 256):
    a, b = t_us(24576, K), t_us(49152, K)
    print("K=%3d ns=%2d  t(24576)=%.1f us  t(49152)=%.1f us  -> %.3f ns/row, fixed %.1f us" % (K, (K + 15) // 16, a, b, (b - a) / 24576 * 1e3, 2 * a - b))
