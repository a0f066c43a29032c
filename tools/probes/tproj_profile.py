"""Where a workgroup of the T projection (gemm_split3_kernel, 64 x 64 tiles, M = 2 560, N = 352, K = 512) spends its life: 100-MHz
stamps at start, after the prologue (index loads, first DMA requests), after the k-loop, after the epilogue (-DJLM_PROFILE build)."""
import ctypes, os, sys, time
os.environ["JLM_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_prof", "libjlm_hip_prof.so")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import _lib
L = _lib.lib()
L.jlm_prof_read_wg_gemm.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
M, N, K = 2560, 352, 512
Af, Bf = torch.randn(3 * M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
A, B, C = torch.zeros_like(Af), torch.zeros_like(Bf), torch.empty((3 * M, N), device=dev)
assert L.jlm_pack_split_f16(Af.data_ptr(), 3 * M, K, K, 1024.0, A.data_ptr(), K, None) == 0
assert L.jlm_pack_split_f16(Bf.data_ptr(), N, K, K, 1024.0, B.data_ptr(), K, None) == 0
rows = (torch.randperm(M, device=dev).to(torch.int32) + 2 * M).contiguous()           # gathered live rows, as in the decode
nd = torch.tensor([M], device=dev, dtype=torch.int32)
f = lambda: L.jlm_gemm_nt_split(A.data_ptr(), K, rows.data_ptr(), B.data_ptr(), K, None, C.data_ptr(), N, rows.data_ptr(), None,
                                2.0 ** -20, M, N, K, nd.data_ptr(), None)
for _ in range(5):
    assert f() == 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): f()
torch.cuda.synchronize()
print("T projection: %.1f us per call (profiled build)" % ((time.perf_counter() - t0) / 100 * 1e6))
buf = (ctypes.c_ulonglong * (4096 * 4))()
assert L.jlm_prof_read_wg_gemm(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
a = a[a[:, 0] > 0][:240]
t0 = a[:, 0].min()
d = (a - t0) / 100.0
print("workgroups %d: start mean %.2f max %.2f us | prologue %.2f | k-loop %.2f (%.3f per 32-wide k-step) | epilogue %.2f | end mean %.2f max %.2f" % (
    len(a), d[:, 0].mean(), d[:, 0].max(), (d[:, 1] - d[:, 0]).mean(), (d[:, 2] - d[:, 1]).mean(), (d[:, 2] - d[:, 1]).mean() / 16,
    (d[:, 3] - d[:, 2]).mean(), d[:, 3].mean(), d[:, 3].max()))
