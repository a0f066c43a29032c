"""per-workgroup timeline of vocab_lse_split_kernel (configs[1] shape) from the -DJLM_PROFILE build"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from jlm_amd import _lib
L = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "..", "build_prof", os.environ.get("JLM_PROF_LIB", "libjlm_hip_prof.so")))
L.jlm_vocab_lse_split.restype = ctypes.c_int
L.jlm_vocab_lse_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [
    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
    ctypes.c_void_p, ctypes.c_void_p]
L.jlm_pack_split_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                 ctypes.c_int, ctypes.c_void_p]
L.jlm_pack_split_f16_col.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.jlm_prof_read_wg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
bounds, widths, R = [0, 12000, 30000, 50000], [200, 100, 50], 2560
n = 3
segs = (_lib.Segment * n)(); ts = (ctypes.c_float * n)(16., 16., 16.); ds = (ctypes.c_float * n)(*([1 / 16384.] * 3))
keep, off = [], 0
for i, k in enumerate(widths):
    kp, k16, nv = (k + 3) // 4 * 4, (k + 15) // 16 * 16, bounds[i + 1] - bounds[i]
    Bm = torch.randn(nv, kp, device=dev) * 0.05; Bs = torch.zeros((nv, k16), device=dev)
    assert L.jlm_pack_split_f16(Bm.data_ptr(), nv, kp, kp, 1024.0, Bs.data_ptr(), k16, None) == 0
    keep += [Bm, Bs]; segs[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, Bs.data_ptr(), k16); off += kp
T, b2 = torch.randn(R, off, device=dev), torch.randn(50000, device=dev) * 0.05
# the bias as a GEMM column (what the decode uses for these widths); BCOL=0: the bias added in the fold
bcol = (ctypes.c_int * n)(*[(w + 3) // 4 * 4 for w in widths]) if os.environ.get("BCOL", "1") != "0" else None
if bcol is not None:
    for i in range(n):
        assert L.jlm_pack_split_f16_col(b2.data_ptr() + 4 * bounds[i], bounds[i + 1] - bounds[i], 1024.0, segs[i].B, segs[i].ldb, bcol[i], None) == 0
part = torch.empty((96, R, 2), device=dev)
f = lambda: L.jlm_vocab_lse_split(segs, ts, ds, bcol, n, b2.data_ptr(), T.data_ptr(), off, None, part.data_ptr(), R, 96, R, None, None)
for _ in range(5): npart = f()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
L.jlm_prof_read_wg(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
a = a[a[:, 0] > 0]                       # 8-wave form: parts x 10 row blocks; 4-wave (JLM_LSE_WAVES=4): parts x 20
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0      # us
print("parts", npart, "WGs", len(a), "kernel span %.1f us" % en.max())
for sgi in range(3):
    m = a[:, 2] == sgi
    print("seg %d: %3d WGs  start %.1f..%.1f us  duration mean %.1f  min %.1f  max %.1f  end max %.1f" % (
        sgi, m.sum(), st[m].min(), st[m].max(), (en - st)[m].mean(), (en - st)[m].min(), (en - st)[m].max(), en[m].max()))
print("shader clock while the workgroups ran: mean %.3f GHz (min %.3f, max %.3f)" % tuple(
    f((a[:, 3] / ((a[:, 1] - a[:, 0]) * 10.0))) for f in (np.mean, np.min, np.max)))
late = st > 5
print("WGs starting later than 5 us:", int(late.sum()), " their starts:", np.sort(st[late])[:12])
