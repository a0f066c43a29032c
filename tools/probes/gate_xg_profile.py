"""Where a workgroup of jlm_lstm_step_xg spends its life (100 MHz wall-clock stamps, -DJLM_PROFILE build of the library:
mkdir -p build_prof && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJLM_PROFILE -o build_prof/libjlm_hip_prof.so jlm_amd/csrc/*.hip).
Per workgroup (waves 0 and 4): index chains, first stage landed, main loop, epilogue."""
import ctypes, os, sys, time
os.environ["JLM_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_prof", "libjlm_hip_prof.so")
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
mode = os.environ.get("GATE_PREV", "random")
if mode == "hot":          # every tile gathers the same 160 state rows, one table row, one cell row: everything L2-resident
    prev = (torch.arange(G, device=dev, dtype=torch.int32) % 160).contiguous()
    word = torch.zeros(G, device=dev, dtype=torch.int32)
elif mode == "ident":      # contiguous state rows instead of a gather
    prev = (torch.arange(G, device=dev, dtype=torch.int32) - 2 * R).clamp(min=0).contiguous()
nd = torch.tensor([R], device=dev, dtype=torch.int32)
f = lambda: L.jlm_lstm_step_xg(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows.data_ptr(), prev.data_ptr(),
                               word.data_ptr(), wt.data_ptr(), xg.data_ptr(), H, 2.0 ** -20, 2.0 ** 14, None, R, nd.data_ptr(), None)
L.jlm_prof_read_gate.argtypes = [ctypes.c_void_p]
for _ in range(5):
    assert f() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    f()
torch.cuda.synchronize()
print("R = %d: %.1f us per call (profiled build)" % (R, (time.perf_counter() - t0) / 50 * 1e6))
buf = (ctypes.c_ulonglong * (2048 * 2 * 8))()
assert L.jlm_prof_read_gate(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 2, 8).astype(np.int64)
nwg = ((R + 159) // 160) * (4 * H // 128)
a = a[:min(nwg, 2048)]
t0 = a[:, :, 0].min()
for w, name in ((0, "wave 0 (3 blocks, requests first)"), (1, "wave 4 (2 blocks, requests last)")):
    x = (a[:, w, :5] - t0) / 100.0          # us since the first workgroup started
    d = np.diff(x, axis=1)
    print(name)
    print("   start          mean %6.2f  max %6.2f us after the first workgroup" % (x[:, 0].mean(), x[:, 0].max()))
    for i, nm in enumerate(("index chains", "table rows + DMA prologue -> first stage landed", "main loop", "epilogue (incl. stores landed)")):
        print("   %-48s mean %6.2f  p10 %6.2f  p90 %6.2f us" % (nm, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    print("   end            mean %6.2f  max %6.2f us" % (x[:, 4].mean(), x[:, 4].max()))
    cyc = (a[:, w, 7] - a[:, w, 6]).astype(np.float64)
    print("   main loop in shader cycles: mean %8.0f = %.3f GHz  (%.0f cycles per k-step at H = 512)" % (
        cyc.mean(), (cyc / (d[:, 2] * 1e3)).mean(), cyc.mean() / 16))
    if a[:, w, 5].max() > 0:
        st = (a[:, w, 5] - a[:, w, 2]) / 100.0
        print("   of the main loop: steady-state k-steps (all but the last 4) %6.2f us, the last 4 %6.2f us" % (
            st.mean(), (d[:, 2] - st).mean()))
