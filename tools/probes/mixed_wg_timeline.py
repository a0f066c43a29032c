"""Workgroup durations of the three-segment jlm_vocab_lse_mixed launch (a -DJLM_WGTIME build: JLM_HIP_LIB=build_prof/libjlm_hip_wgtime.so):
per column (its row tiles' mean) the time from kernel start to its end, and the spread -- what the equal-cost cuts leave on the table."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import _lib
L = _lib.lib()
L.jlm_prof_read_wg_mx.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
R, V = 2560, 50000
widths, bounds = [200, 100, 52], [0, 12000, 30000, 50000]
n = 3
segs = (_lib.Segment * n)()
ts, ds, s8 = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_float * n)()
keep, off = [], 0
b2 = torch.randn(V, device=dev) * 0.05
for i, k in enumerate(widths):
    nb = (k + 2 + 31) // 32
    nv = bounds[i + 1] - bounds[i]
    B = torch.randn(nv, k, device=dev) * 0.05
    dst = torch.zeros((nv, 32 * nb), device=dev)
    assert L.jlm_pack_mixed(B.data_ptr(), nv, k, k, b2.data_ptr() + 4 * bounds[i], 2.0 ** 15, 2.0 ** 15 * 1.4427, 2.0 ** 7, dst.data_ptr(), 32 * nb, None) == 0
    keep += [B, dst]
    segs[i] = _lib.Segment(bounds[i], bounds[i + 1], k, off, dst.data_ptr(), 32 * nb)
    ts[i], ds[i], s8[i] = 2.0 ** 10, 2.0 ** -25, 2.0 ** 7
    off += k
T = torch.randn(R, off, device=dev) * 0.3
ld_tm = L.jlm_mixed_t_stride(segs, n)
Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), device=dev)
assert L.jlm_pack_t_mixed(segs, ts, n, T.data_ptr(), off, None, R, None, Tm.data_ptr(), ld_tm, None) == 0
part = torch.empty((96, R, 2), device=dev)
for _ in range(20):
    ns = L.jlm_vocab_lse_mixed(segs, ds, s8, None, n, Tm.data_ptr(), ld_tm, part.data_ptr(), R, 96, R, None, None)
torch.cuda.synchronize()
buf = np.zeros((1024, 4), dtype=np.uint64)
assert L.jlm_prof_read_wg_mx(buf.ctypes.data) == 0
w = buf[:240].astype(np.float64)
t0 = w[:, 0].min()
end = (w[:, 1] - t0) / 100.0
dur = (w[:, 1] - w[:, 0]) / 100.0
# block b < 240: x = b & 7, jb = b >> 3: column (jb // 10) * 8 + x, row tile jb % 10
b = np.arange(240)
col = ((b >> 3) // 10) * 8 + (b & 7)
print("slices %d; kernel span %.1f us; workgroup duration mean %.1f min %.1f max %.1f us" % (ns, end.max(), dur.mean(), dur.min(), dur.max()))
for c in range(24):
    m = col == c
    print("column %2d: last segment %d, duration mean %.1f us (min %.1f max %.1f)" % (c, int(w[m, 2].max()), dur[m].mean(), dur[m].min(), dur[m].max()))

ghz = w[:, 3] / (dur * 1e3)
print("per XCD (= blockIdx % 8): duration mean, shader clock")
for x in range(8):
    m = (b & 7) == x
    print("   XCD %d: %.1f us  %.3f GHz  cycles %.0f" % (x, dur[m].mean(), ghz[m].mean(), w[m, 3].mean()))
