import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import _lib
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")))
from tests import test_gpu_kernels as tk
L = _lib.lib()
V, bounds, R = 3100, [0, 700, 1900, 3100], 300
widths = [200, 100, 52]
rng = np.random.default_rng(1)
b2_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
mx, ts, ds, s8, Bs, keep, ldt, b2 = tk._mixed_segments(L, rng, V, widths, bounds, b2_np)
torch.cuda.synchronize(); print("mixed segs ok", flush=True)
n = 3
plain = (_lib.Segment * n)()
off = 0
for i in range(n):
    plain[i] = _lib.Segment(bounds[i], bounds[i + 1], widths[i], off, keep[2 * i].data_ptr(), widths[i]); off += widths[i]
sp, sts, sds, bc, keep2 = tk._split_segments(L, plain, None, n, [6] * n, b2)
torch.cuda.synchronize(); print("split segs ok", [bc[i] for i in range(3)], flush=True)
for i in range(n):
    sts[i] = 2.0 ** 10; sds[i] = 2.0 ** -16
mixed = (_lib.Segment * n)()
mixed[0], mixed[1] = mx[0], mx[1]
only = (_lib.Segment * 2)(mx[0], mx[1])
G = R + 9
T = torch.randn(G, ldt, device="cuda") * 0.3
rows = torch.arange(R, dtype=torch.int32, device="cuda")
ld_tm = L.jlm_mixed_t_stride(only, 2)
Tm = torch.zeros((R, ld_tm), dtype=torch.float32, device="cuda")
print("pack", L.jlm_pack_t_mixed(only, (ctypes.c_float * 2)(ts[0], ts[1]), 2, T.data_ptr(), ldt, rows.data_ptr(), R, None, Tm.data_ptr(), ld_tm, None))
torch.cuda.synchronize(); print("pack ok", flush=True)
part = torch.zeros((96, R, 2), dtype=torch.float32, device="cuda")
r0 = L.jlm_vocab_lse_split(sp, sts, sds, bc, n, b2.data_ptr(), T.data_ptr(), ldt, rows.data_ptr(), part.data_ptr(), R, 96, R, None, None)
print("plain split rc", r0, flush=True)
torch.cuda.synchronize(); print("plain split ok", flush=True)
if os.environ.get("HY_NSEG"):
    n = int(os.environ["HY_NSEG"])
r = L.jlm_vocab_lse_hybrid(sp, sts, sds, bc, mixed, ds, s8, n, b2.data_ptr(), T.data_ptr(), ldt, Tm.data_ptr(), ld_tm, rows.data_ptr(), part.data_ptr(), R, 96, R, None, None)
print("hybrid rc", r, flush=True)
torch.cuda.synchronize(); print("hybrid ok", flush=True)
