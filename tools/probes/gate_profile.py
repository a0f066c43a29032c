"""Where a workgroup of the LSTM-step kernel spends its life (clock64 probes, -DJLM_PROFILE build of the library in
build_prof/: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_PROFILE -o build_prof/libjlm_hip_prof.so jlm_amd/csrc/*.hip).
Per wave: prologue (index chain of the gathered rows), main loop, of which waiting at the k-step barrier, epilogue."""
import ctypes, os, sys, time
os.environ["JLM_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_prof", "libjlm_hip_prof.so")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jlm_amd import _lib
L = _lib.lib()
L.jlm_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
H, V, R = 512, 50000, 2560
G = 3 * R
rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev) * scale
hf, c = torch.tanh(rnd(G, H)), rnd(G, H)
wtf = rnd(4 * H, H, scale=0.05)
h, wt = torch.zeros_like(hf), torch.zeros_like(wtf)
assert L.jlm_pack_split_f16(hf.data_ptr(), G, H, H, 2.0 ** 14, h.data_ptr(), H, None) == 0
assert L.jlm_pack_split_f16(wtf.data_ptr(), 4 * H, H, H, 2.0 ** 6, wt.data_ptr(), H, None) == 0
xg = rnd(V, 4 * H)
rows = (torch.arange(R, device=dev, dtype=torch.int32) + 2 * R).contiguous()
prev = torch.randint(0, 2 * R, (G,), device=dev, dtype=torch.int32)
word = torch.randint(0, V, (G,), device=dev, dtype=torch.int32)
nd = torch.tensor([R], device=dev, dtype=torch.int32)
f = lambda: L.jlm_lstm_step_split(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows.data_ptr(), prev.data_ptr(),
                                  word.data_ptr(), None, 0, wt.data_ptr(), None, H, H, 0, 2.0 ** -20, 2.0 ** 14, xg.data_ptr(), R,
                                  nd.data_ptr(), None)
import numpy as np
L.jlm_prof_read_wg_gemm.argtypes = [ctypes.c_void_p]
for _ in range(5):
    assert f() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    f()
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 50 * 1e6
buf = (ctypes.c_ulonglong * (4096 * 4))()
assert L.jlm_prof_read_wg_gemm(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
a = a[a[:, 0] > 0]
k0 = a[:, 0].min()
t = (a - k0) / 100.0                      # us since the first workgroup started
d = lambda i, j: t[:, j] - t[:, i]
print("tile=%s  %.1f us per launch, %d workgroups stamped, kernel span %.1f us" % (os.environ.get("JLM_GATE_TILE", "128"), us, len(a), t[:, 3].max()))
print("  start      mean %5.1f  max %5.1f us after the first workgroup" % (t[:, 0].mean(), t[:, 0].max()))
for name, i, j in (("prologue", 0, 1), ("main loop", 1, 2), ("epilogue", 2, 3), ("lifetime", 0, 3)):
    x = d(i, j)
    print("  %-9s  mean %5.1f  min %5.1f  max %5.1f us" % (name, x.mean(), x.min(), x.max()))
