"""Does anything the mixed vocabulary kernel returns for VALID rows depend on what the packed-row buffer held before (rows past the live
count, earlier frames)?  Packs and runs the same launch twice over different garbage; bit comparison of the valid rows' slices."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import _lib
L = _lib.lib()
dev = torch.device("cuda")
torch.manual_seed(3)
MAXP = int(os.environ.get("MAXP", "96"))
for (V, widths, bounds, R, nlive) in [(2000, [200, 100, 52], [0, 700, 1300, 2000], 384, 300), (2000, [200, 100, 52], [0, 700, 1300, 2000], 48, 48),
                                      (2000, [200, 100, 52], [0, 700, 1300, 2000], 384, 384), (3000, [256], [0, 3000], 384, 300)]:
    n = len(widths)
    segs = (_lib.Segment * n)()
    ts, ds, s8 = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_float * n)()
    keep, off = [], 0
    b2 = torch.randn(V, device=dev) * 0.05
    for i, k in enumerate(widths):
        nb = k // 32 if k % 32 == 0 else (k + 2 + 31) // 32
        nv = bounds[i + 1] - bounds[i]
        Bm = torch.randn(nv, k, device=dev) * 0.05
        dst = torch.zeros((nv, 32 * nb), device=dev)
        assert L.jlm_pack_mixed(Bm.data_ptr(), nv, k, k, b2.data_ptr() + 4 * bounds[i], 2.0 ** 15, 2.0 ** 15 * 1.4427, 2.0 ** 7, dst.data_ptr(), 32 * nb, None) == 0
        keep += [Bm, dst]
        segs[i] = _lib.Segment(bounds[i], bounds[i + 1], k, off, dst.data_ptr(), 32 * nb)
        ts[i], ds[i], s8[i] = 2.0 ** 10, 2.0 ** -25, 2.0 ** 7
        off += k
    b2l = b2 * 1.4426950408889634
    bias2 = b2l.data_ptr() if widths[0] % 32 == 0 else None
    T = torch.randn(R, off, device=dev) * 0.3
    nd = torch.tensor([nlive], device=dev, dtype=torch.int32)
    ld_tm = L.jlm_mixed_t_stride(segs, n)
    outs = []
    for rep in range(2):
        Tm = (torch.randn(((R + 31) // 32 * 32, ld_tm), device=dev) * (1e30 if rep else 1.0))       # garbage (huge the second time)
        part = torch.full((96, R, 2), float(rep), device=dev)
        assert L.jlm_pack_t_mixed(segs, ts, n, T.data_ptr(), off, None, R, nd.data_ptr(), Tm.data_ptr(), ld_tm, None) == 0
        np_ = L.jlm_vocab_lse_mixed(segs, ds, s8, bias2, n, Tm.data_ptr(), ld_tm, part.data_ptr(), R, MAXP, R, nd.data_ptr(), None)
        torch.cuda.synchronize()
        outs.append(part[:np_, :nlive].cpu().numpy().copy())
    # ... and the same launch many times over the same packed rows: any two results that differ mean a race inside the kernel
    ref, n_diff = None, 0
    for rep in range(300):
        part = torch.zeros((96, R, 2), device=dev)
        np_ = L.jlm_vocab_lse_mixed(segs, ds, s8, bias2, n, Tm.data_ptr(), ld_tm, part.data_ptr(), R, MAXP, R, nd.data_ptr(), None)
        torch.cuda.synchronize()
        cur = part[:np_, :nlive].cpu().numpy().tobytes()
        if ref is None:
            ref = cur
        elif cur != ref:
            n_diff += 1
    print("   300 repeats of the launch: %d differ from the first" % n_diff)
    a, b = outs
    bad = np.argwhere(a != b)
    print("V=%d k=%s R=%d live=%d: slices %d; valid rows bit-identical over different garbage: %s%s" % (
        V, widths, R, nlive, a.shape[0], a.tobytes() == b.tobytes(), "" if a.tobytes() == b.tobytes() else "  first differing (slice, row, field): %s of %d" % (bad[0], len(bad))))
