"""Can the small kernels of one batch hide inside the vocabulary kernel of the other?  Stream L runs vocabulary LSE launches
back to back; stream S runs the LSTM-step + T-projection chain of another batch at the same time.  Reports both rates alone
and together, for the LSE forms selected by the environment (JLM_LSE_WAVES / JLM_LSE_NP)."""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jlm_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev) * scale
R, V, H = 2560, 50000, 512
bounds, widths = [0, 12000, 30000, V], [200, 100, 52]
n = 3
segs = (_lib.Segment * n)(); ts, ds, bcol = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_int * n)()
keep, off = [], 0
for i, k in enumerate(widths):
    k16, nv = (k + 15) // 16 * 16, bounds[i + 1] - bounds[i]
    Bm = rnd(nv, k, scale=0.05); Bs = torch.zeros((nv, k16), device=dev)
    assert L.jlm_pack_split_f16(Bm.data_ptr(), nv, k, k, 1024.0, Bs.data_ptr(), k16, None) == 0
    keep += [Bm, Bs]; segs[i] = _lib.Segment(bounds[i], bounds[i + 1], k, off, Bs.data_ptr(), k16); bcol[i] = -1
    ts[i], ds[i] = 16.0, 1.0 / 16384.0; off += k
T, b2 = rnd(R, off), rnd(V, scale=0.05)
part = torch.empty((96, R, 2), device=dev)
nd = torch.tensor([R], device=dev, dtype=torch.int32)
rows = torch.arange(R, device=dev, dtype=torch.int32)
G = 3 * R
hf, c = torch.tanh(rnd(G, H)), rnd(G, H)
wtf, pmf = rnd(4 * H, H, scale=0.05), rnd(352, H, scale=0.05)
h, wt, pm = torch.zeros_like(hf), torch.zeros_like(wtf), torch.zeros_like(pmf)
for src, dst, sc in ((hf, h, 2.0 ** 14), (wtf, wt, 64.0), (pmf, pm, 64.0)):
    assert L.jlm_pack_split_f16(src.data_ptr(), src.shape[0], H, H, sc, dst.data_ptr(), H, None) == 0
xg = rnd(V, 4 * H); T2 = torch.empty((G, 352), device=dev)
rows2 = (torch.arange(R, device=dev, dtype=torch.int32) + 2 * R).contiguous()
prev = torch.randint(0, 2 * R, (G,), device=dev, dtype=torch.int32)
word = torch.randint(0, V, (G,), device=dev, dtype=torch.int32)
sL, sS = torch.cuda.Stream(), torch.cuda.Stream()
def lse(st):
    return L.jlm_vocab_lse_split(segs, ts, ds, bcol, n, b2.data_ptr(), T.data_ptr(), off, rows.data_ptr(), part.data_ptr(), R, 96, R,
                                 nd.data_ptr(), st)
def small(st):
    assert L.jlm_lstm_step_split(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows2.data_ptr(), prev.data_ptr(),
                                 word.data_ptr(), None, 0, wt.data_ptr(), None, H, H, 0, 2.0 ** -20, 2.0 ** 14, xg.data_ptr(), R,
                                 nd.data_ptr(), st) == 0
    assert L.jlm_gemm_nt_split(h.data_ptr(), H, rows2.data_ptr(), pm.data_ptr(), H, None, T2.data_ptr(), 352, rows2.data_ptr(), None,
                               2.0 ** -20, R, 352, H, nd.data_ptr(), st) == 0
def run(do_l, do_s, iters=60):
    torch.cuda.synchronize(); t = time.perf_counter()
    e = []
    for _ in range(iters):
        if do_l: lse(sL.cuda_stream)
        if do_s: small(sS.cuda_stream)
    if do_l:
        sL.synchronize(); tl = time.perf_counter() - t
    else: tl = 0
    if do_s:
        sS.synchronize(); tss = time.perf_counter() - t
    else: tss = 0
    torch.cuda.synchronize()
    return tl / iters * 1e6, tss / iters * 1e6
print("parts", lse(sL.cuda_stream)); run(True, True, 10)
a = run(True, False); b = run(False, True); cc = run(True, True)
print("LSE form: waves=%s np=%s" % (os.environ.get("JLM_LSE_WAVES", "8"), os.environ.get("JLM_LSE_NP", "auto")))
print("  alone:    LSE %.1f us per launch | LSTM step + T projection %.1f us per pair" % (a[0], b[1]))
print("  together: LSE stream done after %.1f us per launch | small-kernel stream after %.1f us per pair" % cc)
