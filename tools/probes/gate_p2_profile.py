"""Where a workgroup of the 128 x 256 persistent LSTM step (gate_p2_kernel, csrc/jlm_gate_p2.hip) spends a tile: shader-clock stamps of waves 0
and 4 at the top of the tile's sixteen k-steps, behind them, when the first epilogue operands are in, behind the cell update; the granted clock
from shader cycles per wall-clock tick (-DJLM_PROFILE build: tools/build_variant.sh PROF -DJLM_PROFILE jlm_gate_p2.hip).  usage: gate_p2_profile.py [rows]"""
import ctypes, os, sys, time
os.environ["JLM_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_prof", os.environ.get("JLM_PROF_LIB", "libjlm_hip_PROF.so"))
os.environ["JLM_GATE_V"] = "4"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from jlm_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
H, V = 512, 50000
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
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
rd = ctypes.CDLL(os.environ["JLM_HIP_LIB"]).jlm_prof_read_gate_p2
rd.argtypes = [ctypes.c_void_p]
for _ in range(5):
    assert f() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    f()
torch.cuda.synchronize()
print("R = %d: %.1f us per call (profiled build)" % (R, (time.perf_counter() - t0) / 50 * 1e6))
buf = (ctypes.c_ulonglong * (256 * 2 * 8 * 5))()
assert rd(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2, 8, 5).astype(np.int64)
ntile = (R + 127) // 128
per = min(8, (ntile + 31) // 32)
wall0 = a[:, :, 0, 4][a[:, :, 0, 4] > 0].min()
for w, name in ((0, "wave 0 (gate wave: requests the gate-matrix pieces)"), (1, "wave 4 (state wave: requests the state pieces)")):
    print(name)
    for t in range(per):
        x = a[:, w, t, :]
        ok = x[:, 3] > x[:, 0]
        if not ok.any():
            continue
        x = x[ok].astype(np.float64)
        ks, wt_, cu = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2]
        line = "  tile %d: k-steps %6.0f cycles (%5.0f per k-step), wait for the epilogue operands %5.0f, cell update %5.0f" % (t, ks.mean(), ks.mean() / 16, wt_.mean(), cu.mean())
        if t + 1 < per:
            nx = a[:, w, t + 1, :][ok].astype(np.float64)
            good = nx[:, 0] > 0
            if good.any():
                cyc, us = (nx[good, 0] - x[good, 0]).mean(), ((nx[good, 4] - x[good, 4]) / 100.0).mean()
                line += "; tile to tile %6.0f cycles = %5.2f us by the wall clock (%.2f GHz)" % (cyc, us, cyc / us / 1e3)
        print(line + "   (%d workgroups; start %.1f us after the first)" % (int(ok.sum()), ((x[:, 4] - wall0) / 100.0).mean()))
