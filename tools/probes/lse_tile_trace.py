"""One wave's clock stamps along one vocabulary tile of the LSE kernel (configs[1] shape), per k-step class
(-DJLM_TILETRACE build: hipcc ... -DJLM_TILETRACE -o build_prof/libjlm_hip_tt.so jlm_amd/csrc/*.hip)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from jlm_amd import _lib
L = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "..", "build_prof", "libjlm_hip_tt.so"))
L.jlm_vocab_lse_split.restype = ctypes.c_int
L.jlm_vocab_lse_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [
    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
    ctypes.c_void_p, ctypes.c_void_p]
L.jlm_pack_split_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                 ctypes.c_int, ctypes.c_void_p]
L.jlm_pack_split_f16_col.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.jlm_tile_trace_read.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
bounds, widths, R = [0, 12000, 30000, 50000], [200, 100, 52], 2560
n = 3
segs = (_lib.Segment * n)(); ts = (ctypes.c_float * n)(16., 16., 16.); ds = (ctypes.c_float * n)(*([1 / 16384.] * 3))
bcol = (ctypes.c_int * n)()
keep, off = [], 0
T0 = None
b2 = torch.randn(50000, device=dev) * 0.05
for i, k in enumerate(widths):
    k16, nv = (k + 15) // 16 * 16, bounds[i + 1] - bounds[i]
    Bm = torch.randn(nv, k, device=dev) * 0.05; Bs = torch.zeros((nv, k16), device=dev)
    assert L.jlm_pack_split_f16(Bm.data_ptr(), nv, k, k, 1024.0, Bs.data_ptr(), k16, None) == 0
    bcol[i] = k
    assert L.jlm_pack_split_f16_col(b2.data_ptr() + 4 * bounds[i], nv, 1024.0, Bs.data_ptr(), k16, k, None) == 0
    keep += [Bm, Bs]; segs[i] = _lib.Segment(bounds[i], bounds[i + 1], k, off, Bs.data_ptr(), k16); off += k
T = torch.randn(R, off, device=dev)
part = torch.empty((96, R, 2), device=dev)
f = lambda: L.jlm_vocab_lse_split(segs, ts, ds, bcol, n, b2.data_ptr(), T.data_ptr(), off, None, part.data_ptr(), R, 96, R, None, None)
for _ in range(5): npart = f()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * 128))()
assert L.jlm_tile_trace_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(3, 128).astype(np.int64)
names = {1: "tile start", 2: "chunk's MFMAs issued", 3: "barrier passed (waiting for the other waves)", 4: "fold done", 5: "next chunk's DMA landed (vmcnt 0)"}
for cls, k in enumerate(widths):
    cnt = int(a[cls, 0])
    ev = [(int(a[cls, 1 + 2 * i]), int(a[cls, 2 + 2 * i])) for i in range(cnt)]
    if not ev: continue
    t0 = ev[0][1]
    print("k = %d (%d k-steps): tile %d cycles" % (k, (k + 16) // 16, ev[-1][1] - t0))
    last = t0
    for tag, clk in ev[1:]:
        nm = names.get(tag, "k-step %d MFMAs issued" % (tag - 10))
        print("   +%5d  (%6d)  %s" % (clk - last, clk - t0, nm)); last = clk
