"""Where a workgroup of the W-stationary persistent LSTM step (gate_ws_kernel, csrc/jlm_gate_ws.hip) spends its life: 100 MHz stamps of
wave 0 (-DJLM_PROFILE build: build_prof/libjlm_hip_prof.so, tools/gpu_gate_ws.sh builds it).  usage: gate_ws_profile.py [rows]"""
import ctypes, os, sys, time
os.environ["JLM_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_prof", os.environ.get("JLM_PROF_LIB", "libjlm_hip_prof.so"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from jlm_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
H, V = 512, 50000
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
G = 3 * R
rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev) * scale
hf, c = torch.tanh(rnd(G, H)), rnd(G, H)
wtf = rnd(4 * H, H, scale=0.05)
h, wt = torch.zeros_like(hf), torch.zeros_like(wtf)
assert L.jlm_pack_split_f16(hf.data_ptr(), G, H, H, 2.0 ** 14, h.data_ptr(), H, None) == 0
assert L.jlm_pack_split_f16(wtf.data_ptr(), 4 * H, H, H, 2.0 ** 6, wt.data_ptr(), H, None) == 0
xg = rnd(V, 4 * H, scale=2.0 ** 20)
rows = (torch.arange(R, device=dev, dtype=torch.int32) + 2 * R).contiguous()
word = torch.randint(0, V, (G,), device=dev, dtype=torch.int32)
prev = torch.randint(0, 2 * R, (G,), device=dev, dtype=torch.int32)
nd = torch.tensor([R], device=dev, dtype=torch.int32)
f = lambda: L.jlm_lstm_step_xg(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows.data_ptr(), prev.data_ptr(),
                               word.data_ptr(), wt.data_ptr(), xg.data_ptr(), H, 2.0 ** -20, 2.0 ** 14, None, R, nd.data_ptr(), None)
rd = ctypes.CDLL(os.environ["JLM_HIP_LIB"]).jlm_prof_read_gate_ws
rd.argtypes = [ctypes.c_void_p]
for _ in range(5):
    assert f() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    f()
torch.cuda.synchronize()
print("R = %d: %.1f us per call (profiled build), JLM_GATE_WS_L=%s CX=%s" % (R, (time.perf_counter() - t0) / 50 * 1e6, os.environ.get("JLM_GATE_WS_L"), os.environ.get("JLM_GATE_WS_CX")))
buf = (ctypes.c_ulonglong * (256 * 32))()
assert rd(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 32).astype(np.int64)
ntile = (R + 159) // 160
per = (ntile + 15) // 16
act = a[a[:, 3] > 0] if per else a
t00 = act[:, 0].min()
x = (act[:, :30] - t00) / 100.0
print("workgroups with work: %d; tiles per workgroup up to %d" % (len(act), per))
print("  start after the first workgroup     mean %6.2f max %6.2f us" % (x[:, 0].mean(), x[:, 0].max()))
print("  index loads                         mean %6.2f us" % (x[:, 1] - x[:, 0]).mean())
print("  first stage (+ its gate fragments)  mean %6.2f us" % (x[:, 2] - x[:, 1]).mean())
cyc = (act[:, 31] - act[:, 30]).astype(np.float64)
d0 = x[:, 3] - x[:, 2]
print("  tile 0: k-steps %6.2f us = %6.0f cycles per k-step at %.2f GHz; cell update %6.2f us" % (
    d0.mean(), cyc.mean() / 16, (cyc / (d0 * 1e3)).mean(), (x[:, 4] - x[:, 3]).mean()))
for t in range(1, min(per, 13)):
    ok = act[:, 3 + 2 * t] > 0
    if ok.any():
        print("  tile %d: k-steps %6.2f us, cell update %6.2f us   (%d workgroups)" % (
            t, (x[ok, 3 + 2 * t] - x[ok, 2 + 2 * t]).mean(), (x[ok, 4 + 2 * t] - x[ok, 3 + 2 * t]).mean(), int(ok.sum())))
last = np.array([r[[i for i in range(4, 30, 2) if r[i] > 0][-1]] for r in (act[:, :30])])
print("  end                                 mean %6.2f max %6.2f us" % (((last - t00) / 100.0).mean(), ((last - t00) / 100.0).max()))
