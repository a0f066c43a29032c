"""Per-kernel micro-benchmark on the MI355X box (HIP events around single launches,
random data, shapes of BASELINE configs[1]/[2]).  Usage: python tools/kbench.py [filter]"""
import os
import sys
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from jlm_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
flt = sys.argv[1] if len(sys.argv) > 1 else ""
if os.environ.get("KBENCH_DUMP"):
    torch.manual_seed(1234)
PEAK = 157.3


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(t)), float(t.min())


def report(name, flops, ms):
    med, mn = ms
    print("%-44s median %8.1f us  min %8.1f us  %6.1f TF/s (%4.1f%% of f32 MFMA peak)" % (
        name, med * 1e3, mn * 1e3, flops / (med * 1e-3) / 1e12, 100 * flops / (med * 1e-3) / 1e12 / PEAK))


def rnd(*shape, scale=1.0):
    if os.environ.get("KBENCH_ZERO"):        # all-zero operands: the same instruction stream at the lowest switching power (is a kernel power-bound?)
        return torch.zeros(*shape, device=dev)
    return (torch.randn(*shape, device=dev) * scale).contiguous()


def bench_lse(V, K, R, tag):
    if flt and flt not in "lse":
        return
    Kp = (K + 3) // 4 * 4
    B, T, bias = rnd(V, Kp, scale=0.05), rnd(R, Kp), rnd(V, scale=0.05)
    ntile = (V + 127) // 128
    part = torch.empty((ntile, R, 2), device=dev)
    lse = torch.empty(R, device=dev, dtype=torch.float64)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    rows = torch.arange(R, device=dev, dtype=torch.int32)
    f = lambda: L.jlm_vocab_lse_partials(B.data_ptr(), Kp, V, Kp, T.data_ptr(), Kp, rows.data_ptr(), bias.data_ptr(),
                                         part.data_ptr(), R, 0, R, nd.data_ptr(), st)
    report("vocab_lse %s V=%d K=%d R=%d" % (tag, V, K, R), 2.0 * V * K * R, timeit(f))
    g = lambda: L.jlm_lse_combine(part.data_ptr(), R, ntile, rows.data_ptr(), lse.data_ptr(), R, nd.data_ptr(), st)
    report("lse_combine tiles=%d R=%d" % (ntile, R), 1.0, timeit(g))


def bench_lse_stat(V, widths, R, tag):
    if flt and flt not in "lse":
        return
    if os.environ.get("KBENCH_ONLY") and os.environ["KBENCH_ONLY"] not in "stat " + tag:
        return
    bounds = [0, 12000, 30000, V] if len(widths) == 3 else [0, V]
    segs = (_lib.Segment * len(widths))()
    keep, off, flops = [], 0, 0.0
    for i, k in enumerate(widths):
        kp = (k + 3) // 4 * 4
        nv = bounds[i + 1] - bounds[i]
        Bm = rnd(nv, kp, scale=0.05)
        keep.append(Bm)
        segs[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, Bm.data_ptr(), kp)
        off += kp
        flops += 2.0 * k * nv * R
    T, b2 = rnd(R, off), rnd(V, scale=0.05)
    part = torch.empty((96, R, 2), device=dev)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    rows = torch.arange(R, device=dev, dtype=torch.int32)
    f = lambda: L.jlm_vocab_lse_stationary(segs, len(widths), b2.data_ptr(), T.data_ptr(), off, rows.data_ptr(),
                                           part.data_ptr(), R, 96, R, nd.data_ptr(), st)
    print("parts:", f())
    report("vocab_lse_stationary %s V=%d k=%s R=%d" % (tag, V, widths, R), flops, timeit(f))


def bench_lse_split(V, widths, R, tag, bias_col=False):
    if flt and flt not in "lse":
        return
    if os.environ.get("KBENCH_ONLY") and os.environ["KBENCH_ONLY"] not in "split " + tag:
        return
    import ctypes
    bounds = [0, 12000, 30000, V] if len(widths) == 3 else [0, V]
    n = len(widths)
    segs = (_lib.Segment * n)()
    ts, ds, bcol = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_int * n)()
    keep, off, flops = [], 0, 0.0
    for i, k in enumerate(widths):
        kp = (k + 3) // 4 * 4
        k16 = (k + 15) // 16 * 16
        nv = bounds[i + 1] - bounds[i]
        Bm = rnd(nv, kp, scale=0.05)
        Bs = torch.zeros((nv, k16), device=dev)
        assert L.jlm_pack_split_f16(Bm.data_ptr(), nv, kp, kp, 1024.0, Bs.data_ptr(), k16, st) == 0
        keep += [Bm, Bs]
        segs[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, Bs.data_ptr(), k16)
        bcol[i] = kp if (bias_col and kp % 16) else -1
        ts[i], ds[i] = 16.0, 1.0 / (16.0 * 1024.0)
        off += kp
        flops += 2.0 * k * nv * R
    T, b2 = rnd(R, off), rnd(V, scale=0.05)
    part = torch.empty((96, R, 2), device=dev)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    rows = torch.arange(R, device=dev, dtype=torch.int32)
    if bias_col:
        for i in range(n):
            if bcol[i] >= 0:
                assert L.jlm_pack_split_f16_col(b2.data_ptr() + 4 * bounds[i], bounds[i + 1] - bounds[i], 1024.0,
                                                segs[i].B, segs[i].ldb, bcol[i], st) == 0
    f = lambda: L.jlm_vocab_lse_split(segs, ts, ds, bcol, n, b2.data_ptr(), T.data_ptr(), off, rows.data_ptr(),
                                      part.data_ptr(), R, 96, R, nd.data_ptr(), st)
    print("parts:", f())
    report("vocab_lse_split%s %s V=%d k=%s R=%d" % ("+bcol " if bias_col else "      ", tag, V, widths, R), flops, timeit(f))


def bench_lse_mixed(V, widths, R, tag, fmt="int8", reps=1):
    """jlm_vocab_lse_mixed: f16 hi.hi + the cross terms on int8 planes, or (fmt "mx6", ABI 11) on FP6 planes with block scales"""
    if fmt == "mx6" and max(widths) > 256:
        return
    if flt and flt not in "lse":
        return
    if os.environ.get("KBENCH_ONLY") and os.environ["KBENCH_ONLY"] not in "mixed split " + tag:
        return
    import ctypes
    bounds = [0, 12000, 30000, V] if len(widths) == 3 else [0, V]
    n = len(widths)
    segs = (_lib.Segment * n)()
    ts, ds, s8 = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_float * n)()
    keep, off, flops = [], 0, 0.0
    b2 = rnd(V, scale=0.05)
    for i, k in enumerate(widths):
        kp = (k + 3) // 4 * 4
        nb = kp // 32 if kp % 32 == 0 else (kp + 2 + 31) // 32       # (k a multiple of 32: no bias columns, biases from bias2)
        nv = bounds[i + 1] - bounds[i]
        Bm = rnd(nv, kp, scale=0.05)
        dst = torch.zeros((nv, 32 * nb), device=dev)
        sb = 0.0 if fmt == "mx6" else 2.0 ** 7
        assert L.jlm_pack_mixed(Bm.data_ptr(), nv, kp, kp, b2.data_ptr() + 4 * bounds[i], 2.0 ** 15, 2.0 ** 15 * 1.4427, sb,
                                dst.data_ptr(), 32 * nb, st) == 0
        keep += [Bm, dst]
        segs[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, dst.data_ptr(), 32 * nb)
        ts[i], ds[i], s8[i] = 2.0 ** 10, 2.0 ** -25, sb
        off += kp
        flops += 2.0 * k * nv * R
    b2l = b2 * 1.4426950408889634
    bias2 = b2l.data_ptr() if widths[0] % 32 == 0 else None
    T = rnd(R, off)
    part = torch.empty((96, R, 2), device=dev)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    rows = torch.arange(R, device=dev, dtype=torch.int32)
    ld_tm = L.jlm_mixed_t_stride(segs, n)
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), device=dev)
    pack_t = L.jlm_pack_t_mixed6 if fmt == "mx6" else L.jlm_pack_t_mixed
    g = lambda: pack_t(segs, ts, n, T.data_ptr(), off, rows.data_ptr(), R, nd.data_ptr(), Tm.data_ptr(), ld_tm, st)
    assert g() == 0
    f = lambda: L.jlm_vocab_lse_mixed(segs, ds, s8, bias2, n, Tm.data_ptr(), ld_tm, part.data_ptr(), R, 96, R, nd.data_ptr(), st)
    np_ = f()
    print("parts:", np_)
    if os.environ.get("KBENCH_DUMP"):        # bit comparison of two builds: same seed, the partial (max, sum) pairs to a file
        torch.cuda.synchronize()
        np.save("%s.%s.k%s.npy" % (os.environ["KBENCH_DUMP"], tag, "_".join(map(str, widths))), part[:np_].cpu().numpy())
    for _ in range(reps):
        report("vocab_lse_mixed %-5s %s V=%d k=%s R=%d" % (fmt, tag, V, widths, R), flops, timeit(f))
    report("pack_t_mixed    %-5s %s R=%d" % (fmt, tag, R), 1.0, timeit(g))


def bench_lse_hybrid(V, widths, R, tag, mixed_set=(0, 1), heads=None):
    """jlm_vocab_lse_hybrid: the segments of mixed_set on mixed rows, the rest on split rows with a bias column; heads: the leading words
    of a mixed segment that stay on split rows (ABI 10)"""
    if flt and flt not in "lse":
        return
    if os.environ.get("KBENCH_ONLY") and os.environ["KBENCH_ONLY"] not in "hybrid mixed split " + tag:
        return
    import ctypes
    bounds = [0, 12000, 30000, V]
    n = len(widths)
    sp, mx = (_lib.Segment * n)(), (_lib.Segment * n)()
    ts, ds, bcol = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_int * n)()
    mts, mds, ms8 = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_float * n)()
    keep, off, flops = [], 0, 0.0
    b2 = rnd(V, scale=0.05)
    for i, k in enumerate(widths):
        kp = (k + 3) // 4 * 4
        k16 = (k + 15) // 16 * 16
        nv = bounds[i + 1] - bounds[i]
        Bm = rnd(nv, kp, scale=0.05)
        Bs = torch.zeros((nv, k16), device=dev)
        assert L.jlm_pack_split_f16(Bm.data_ptr(), nv, kp, kp, 1024.0, Bs.data_ptr(), k16, st) == 0
        keep += [Bm, Bs]
        sp[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, Bs.data_ptr(), k16)
        bcol[i] = kp if kp % 16 else -1
        if bcol[i] >= 0:
            assert L.jlm_pack_split_f16_col(b2.data_ptr() + 4 * bounds[i], nv, 1024.0, sp[i].B, sp[i].ldb, bcol[i], st) == 0
        ts[i], ds[i] = 16.0, 1.0 / (16.0 * 1024.0)
        if i in mixed_set:
            nb = (kp + 2 + 31) // 32
            dst = torch.zeros((nv, 32 * nb), device=dev)
            assert L.jlm_pack_mixed(Bm.data_ptr(), nv, kp, kp, b2.data_ptr() + 4 * bounds[i], 2.0 ** 15, 2.0 ** 15 * 1.4427, 2.0 ** 7,
                                    dst.data_ptr(), 32 * nb, st) == 0
            keep.append(dst)
            mx[i] = _lib.Segment(bounds[i], bounds[i + 1], kp, off, dst.data_ptr(), 32 * nb)
            mts[i], mds[i], ms8[i] = 2.0 ** 10, 2.0 ** -25, 2.0 ** 7
        off += kp
        flops += 2.0 * k * nv * R
    T = rnd(R, off)
    part = torch.empty((96, R, 2), device=dev)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    rows = torch.arange(R, device=dev, dtype=torch.int32)
    n_mixed = len(mixed_set)
    only = (_lib.Segment * n_mixed)(*[mx[i] for i in mixed_set])
    only_ts = (ctypes.c_float * n_mixed)(*[mts[i] for i in mixed_set])
    ld_tm = L.jlm_mixed_t_stride(only, n_mixed)
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), device=dev)
    g = lambda: L.jlm_pack_t_mixed(only, only_ts, n_mixed, T.data_ptr(), off, rows.data_ptr(), R, nd.data_ptr(), Tm.data_ptr(), ld_tm, st)
    assert g() == 0
    hs = (ctypes.c_int * n)(*heads) if heads else None
    f = lambda: L.jlm_vocab_lse_hybrid(sp, ts, ds, bcol, mx, mds, ms8, hs, n, b2.data_ptr(), T.data_ptr(), off, Tm.data_ptr(), ld_tm,
                                       rows.data_ptr(), part.data_ptr(), R, 96, R, nd.data_ptr(), st)
    print("parts:", f())
    what = "mixed %s heads %s" % (list(mixed_set), heads)
    report("vocab_lse_hybrid     %s V=%d k=%s R=%d %s" % (tag, V, widths, R, what), flops, timeit(f))
    report("pack_t_mixed (%d seg) %s R=%d" % (n_mixed, tag, R), 1.0, timeit(g))
    report("pack_t + hybrid      %s R=%d %s" % (tag, R, what), flops, timeit(lambda: (g(), f())))


def bench_gate(H, E, R):
    if flt and flt not in "gate":
        return
    V = 50000
    kpad = (H + E + 31) // 32 * 32
    G = 3 * R
    h, c = rnd(G, H, scale=0.3), rnd(G, H, scale=0.3)
    emb = rnd(V, E, scale=0.05)
    wt = rnd(4 * H, kpad, scale=0.05)
    bias = rnd(4 * H, scale=0.05)
    rows = (torch.arange(R, device=dev, dtype=torch.int32) + 2 * R).contiguous()
    prev = torch.randint(0, 2 * R, (G,), device=dev, dtype=torch.int32)
    word = torch.randint(0, V, (G,), device=dev, dtype=torch.int32)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    f = lambda: L.jlm_lstm_step(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows.data_ptr(), prev.data_ptr(),
                                word.data_ptr(), emb.data_ptr(), E, wt.data_ptr(), bias.data_ptr(), kpad, H, E, R,
                                nd.data_ptr(), st)
    report("lstm_step H=%d E=%d R=%d" % (H, E, R), 2.0 * (H + E) * 4 * H * R, timeit(f))


def bench_gate_xg(H, R):
    if flt and flt not in "gate":
        return
    V, G = 50000, 3 * R
    hf, c = torch.tanh(rnd(G, H)), rnd(G, H)
    wtf = rnd(4 * H, H, scale=0.05)
    h, wt = torch.zeros_like(hf), torch.zeros_like(wtf)
    assert L.jlm_pack_split_f16(hf.data_ptr(), G, H, H, 2.0 ** 14, h.data_ptr(), H, st) == 0
    assert L.jlm_pack_split_f16(wtf.data_ptr(), 4 * H, H, H, 2.0 ** 6, wt.data_ptr(), H, st) == 0
    xg = rnd(V, 4 * H, scale=2.0 ** 20)
    rows = (torch.arange(R, device=dev, dtype=torch.int32) + 2 * R).contiguous()
    prev = torch.randint(0, 2 * R, (G,), device=dev, dtype=torch.int32)
    word = torch.randint(0, V, (G,), device=dev, dtype=torch.int32)
    if os.environ.get("KBENCH_WORDS") == "zipf":      # word ids drawn ~ 1 / rank (a frequency-sorted lexicon): table lines repeat, as in a decode
        w = 1.0 / torch.arange(1, V + 1, device=dev, dtype=torch.float64)
        word = torch.multinomial(w / w.sum(), G, replacement=True).to(torch.int32)
    nd = torch.tensor([R], device=dev, dtype=torch.int32)
    f = lambda: L.jlm_lstm_step_xg(h.data_ptr(), c.data_ptr(), H, h.data_ptr(), c.data_ptr(), rows.data_ptr(), prev.data_ptr(),
                                   word.data_ptr(), wt.data_ptr(), xg.data_ptr(), H, 2.0 ** -20, 2.0 ** 14, None, R, nd.data_ptr(), st)
    assert f() == 0
    med, mn = timeit(f)
    ex = 3 * 2.0 * H * 4 * H * R
    print("lstm_step_xg H=%d R=%d                        median %8.1f us  min %8.1f us  executed %6.1f TF/s = %4.1f%% of dense f16 peak "
          "(counted 2(H+200)4H: %5.1f TF)" % (H, R, med * 1e3, mn * 1e3, ex / (med * 1e-3) / 1e12,
                                               100 * ex / (med * 1e-3) / 1e12 / 2516.6, 2.0 * (H + 200) * 4 * H * R / (med * 1e-3) / 1e12))


def bench_gemm_split(M, N, K, tag):
    if flt and flt not in "gemm":
        return
    Af, Bf, C = rnd(M, K), rnd(N, K), torch.empty((M, N), device=dev)
    A, B = torch.zeros_like(Af), torch.zeros_like(Bf)
    assert L.jlm_pack_split_f16(Af.data_ptr(), M, K, K, 1024.0, A.data_ptr(), K, st) == 0
    assert L.jlm_pack_split_f16(Bf.data_ptr(), N, K, K, 1024.0, B.data_ptr(), K, st) == 0
    f = lambda: L.jlm_gemm_nt_split(A.data_ptr(), K, None, B.data_ptr(), K, None, C.data_ptr(), N, None, None, 2.0 ** -20,
                                    M, N, K, None, st)
    assert f() == 0
    report("gemm_nt_split %s M=%d N=%d K=%d" % (tag, M, N, K), 2.0 * M * N * K, timeit(f))


def bench_gemm(M, N, K, tag):
    if flt and flt not in "gemm":
        return
    A, B, C = rnd(M, K), rnd(N, K), torch.empty((M, N), device=dev)
    f = lambda: L.jlm_gemm_nt(A.data_ptr(), K, None, B.data_ptr(), K, None, C.data_ptr(), N, None, None, M, N, K, None, st)
    report("gemm_nt %s M=%d N=%d K=%d" % (tag, M, N, K), 2.0 * M * N * K, timeit(f))


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    if os.environ.get("KBENCH_MX6"):          # round 6: the two formats of the cross-term planes side by side, interleaved
        fmts = tuple(os.environ.get("KBENCH_FMTS", "int8,mx6").split(","))
        for rep in range(int(os.environ["KBENCH_MX6"])):
            for fmt in fmts:
                bench_lse_mixed(50000, [200, 100, 50], 2560, "dsoftmax*", fmt)
                for V1, k1 in ((12000, 200), (18000, 100), (20000, 50)):
                    bench_lse_mixed(V1, [k1], 2560, "seg-k%d" % k1, fmt)
                bench_lse_mixed(50000, [256], 2560, "tied50k", fmt)
        if not os.environ.get("KBENCH_NO_BIG"):
            for fmt in fmts:
                bench_lse_mixed(100000, [256], 20480, "tied100k-b20", fmt)
        sys.exit(0)
    for R in (2560,):
        bench_gate(512, 200, R)
        bench_gate_xg(512, R)
        bench_gate(512, 256, R)
        bench_lse_stat(50000, [200, 100, 50], R, "dsoftmax*")
        bench_lse_split(50000, [200, 100, 50], R, "dsoftmax*")
        bench_lse_split(50000, [200, 100, 50], R, "dsoftmax*", bias_col=True)
        bench_lse_mixed(50000, [200, 100, 50], R, "dsoftmax*")
        bench_lse_hybrid(50000, [200, 100, 50], R, "dsoftmax*")
        bench_lse_hybrid(50000, [200, 100, 50], R, "dsoftmax*", mixed_set=(0, 1, 2))
        bench_lse_hybrid(50000, [200, 100, 50], R, "dsoftmax*", mixed_set=(0, 1, 2), heads=[2048, 0, 0])
        bench_lse_hybrid(50000, [200, 100, 50], R, "dsoftmax*", mixed_set=(0, 1, 2), heads=[8192, 0, 0])
        bench_lse_hybrid(50000, [200, 100, 50], R, "dsoftmax*", mixed_set=(1, 2))
        bench_lse_hybrid(50000, [200, 100, 50], R, "dsoftmax*", mixed_set=(1,))
        if os.environ.get("KBENCH_SEGS"):
            for V1, k1 in ((12000, 200), (18000, 100), (20000, 50)):
                bench_lse_split(V1, [k1], R, "seg-k%d" % k1, bias_col=True)
                bench_lse_mixed(V1, [k1], R, "seg-k%d" % k1)
        bench_lse_stat(50000, [256], R, "tied50k")
        bench_lse_split(50000, [256], R, "tied50k")
        bench_lse_mixed(50000, [252], R, "tied50k")          # (the widest contraction with room for the bias columns: nb = 8)
        bench_lse_mixed(50000, [256], R, "tied50k")          # the tied shape: biases from bias2
        bench_lse_mixed(50000, [512], R, "untied50k")        # k = 512: the wide kernel's one-row-set form (an untied model's vocabulary matrix)
        bench_lse(12000, 200, R, "seg0")
        bench_lse(18000, 100, R, "seg1")
        bench_lse(20000, 50, R, "seg2")
        bench_lse(50000, 256, R, "tied50k")
        bench_gemm(R, 352, 512, "T")
        bench_gemm_split(R, 352, 512, "T")
        bench_gemm(R, 200, 512, "PM")
        bench_gemm(R, 100, 200, "VT1")
        bench_gemm(R, 52, 200, "VT2")
    bench_gate_xg(512, 5120)
    bench_gate_xg(512, 10240)
    bench_gate_xg(512, 20480)
    bench_gate(512, 256, 20480)
    bench_lse(100000, 256, 20480, "tied100k-b20")
    bench_lse_stat(100000, [256], 20480, "tied100k-b20")
    bench_lse_split(100000, [256], 20480, "tied100k-b20")
    bench_lse_mixed(100000, [256], 20480, "tied100k-b20")
    bench_gemm(4096, 4096, 4096, "square")
    bench_gemm_split(4096, 4096, 4096, "square")
