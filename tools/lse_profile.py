"""Cycle accounting of vocab_lse_stationary_kernel (per-wave clock64 probes).

Needs the -DJLM_PROFILE build of the HIP library (tools/lse_profile.sh makes it in
build_prof/); prints, per workload, the share of a wave's lifetime spent in the
prologue (T fragments), waiting at k-step barriers, in the fold, and the rest
(MFMA issue + fragment reads)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from jlm_amd import _lib  # noqa: E402

L = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "build_prof", "libjlm_hip_prof.so"))
L.jlm_vocab_lse_stationary.restype = ctypes.c_int
L.jlm_vocab_lse_stationary.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p]
L.jlm_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.jlm_prof_read_split.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.jlm_vocab_lse_split.restype = ctypes.c_int
L.jlm_vocab_lse_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [
    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
    ctypes.c_void_p, ctypes.c_void_p]
L.jlm_pack_split_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                 ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")


def run(bounds, widths, R, tag, iters=20, split=False):
    segs = (_lib.Segment * len(widths))()
    ts = (ctypes.c_float * len(widths))(*([16.0] * len(widths)))
    ds = (ctypes.c_float * len(widths))(*([1.0 / (16 * 1024)] * len(widths)))
    keep, off, flops = [], 0, 0.0
    for i, k in enumerate(widths):
        kp = (k + 3) // 4 * 4
        nv = bounds[i + 1] - bounds[i]
        Bm = (torch.randn(nv, kp, device=dev) * 0.05)
        keep.append(Bm)
        if split:
            k16 = (k + 15) // 16 * 16
            Bs = torch.zeros((nv, k16), device=dev)
            assert L.jlm_pack_split_f16(Bm.data_ptr(), nv, kp, kp, 1024.0, Bs.data_ptr(), k16, None) == 0
            keep.append(Bs)
            segs[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, Bs.data_ptr(), k16)
        else:
            segs[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, Bm.data_ptr(), kp)
        off += kp
        flops += 2.0 * k * nv * R
    T, b2 = torch.randn(R, off, device=dev), torch.randn(bounds[-1], device=dev) * 0.05
    part = torch.empty((96, R, 2), device=dev)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    rows = torch.arange(R, device=dev, dtype=torch.int32)
    if split:
        f = lambda: L.jlm_vocab_lse_split(segs, ts, ds, None, len(widths), b2.data_ptr(), T.data_ptr(), off, rows.data_ptr(),
                                          part.data_ptr(), R, 96, R, nd.data_ptr(), None)
    else:
        f = lambda: L.jlm_vocab_lse_stationary(segs, len(widths), b2.data_ptr(), T.data_ptr(), off, rows.data_ptr(),
                                               part.data_ptr(), R, 96, R, nd.data_ptr(), None)
    read = L.jlm_prof_read_split if split else L.jlm_prof_read
    for _ in range(3):
        n = f()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    read(out, 1)
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / iters * 1e6
    read(out, 1)
    tot, pro, first, bar, fold, nw = [float(out[i]) for i in range(6)]
    wall = float(out[6])
    mf = tot - pro - first - bar - fold
    tag = ("split " if split else "f32   ") + tag
    print("   shader clock while the waves ran: %.2f GHz" % (tot / max(wall, 1.0) * 0.1))
    print("%-40s parts=%d  %.0f us (%.1f TF)  wave life %.0f cyc: T-frags %.1f%%  first chunk %.1f%%  barrier %.1f%%  fold %.1f%%  mfma+frag %.1f%%"
          % (tag, n, us, flops / us / 1e6, tot / nw, 100 * pro / tot, 100 * first / tot, 100 * bar / tot, 100 * fold / tot,
             100 * mf / tot))


for sp in (False, True):
    run([0, 12000, 30000, 50000], [200, 100, 50], 2560, "dsoftmax* R=2560", split=sp)
    run([0, 12000], [200], 2560, "seg0 12000x200 R=2560", split=sp)
    run([0, 18000], [100], 2560, "seg1 18000x100 R=2560", split=sp)
    run([0, 20000], [50], 2560, "seg2 20000x50 R=2560", split=sp)
    run([0, 50000], [256], 2560, "tied 50000x256 R=2560", split=sp)
    run([0, 100000], [256], 20480, "tied 100000x256 R=20480", iters=5, split=sp)
