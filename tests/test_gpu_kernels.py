"""GPU: every entry point of libjlm_hip.so against the numpy restatement of its
contract (tests/fake_hip.py), same inputs, called through the C ABI."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from jlm_amd import _lib            # noqa: E402
from tests.fake_hip import FakeLib  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    lib = _lib.lib()
    buf = ctypes.create_string_buffer(128)
    assert lib.jlm_device_arch(0, buf, 128) == 0
    assert buf.value.decode().startswith("gfx950"), buf.value
    return lib


FK = FakeLib()


def _pair(a):
    """(cpu tensor, gpu clone)"""
    t = torch.as_tensor(a).contiguous()
    return t, t.cuda()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _pad_rows(rng, n, k, ld, scale=1.0):
    a = np.zeros((n, ld), dtype=np.float32)
    a[:, :k] = rng.standard_normal((n, k)).astype(np.float32) * scale
    return a


@pytest.mark.parametrize("M,N,K,maps", [(77, 203, 100, False), (300, 64, 512, True), (1500, 6000, 256, True),
                                        (2560, 256, 512, True), (1, 8, 4, False)])
def test_gemm_nt(L, M, N, K, maps):
    rng = np.random.default_rng(M + N + K)
    lda, ldb, ldc = K + 4, K, N + 8
    nA, nB, nC = M + 13, N + 5, M + 9
    A, Ag = _pair(_pad_rows(rng, nA, K, lda))
    B, Bg = _pair(_pad_rows(rng, nB, K, ldb))
    bias, biasg = _pair(rng.standard_normal(N).astype(np.float32))
    C, Cg = _pair(np.full((nC, ldc), 7.0, dtype=np.float32))
    if maps:
        am, amg = _pair(rng.permutation(nA)[:M].astype(np.int32))
        bm, bmg = _pair(rng.permutation(nB)[:N].astype(np.int32))
        cm, cmg = _pair(rng.permutation(nC)[:M].astype(np.int32))
        md, mdg = _pair(np.array([M - 3], dtype=np.int32))
        pc = (am.data_ptr(), bm.data_ptr(), cm.data_ptr(), md.data_ptr())
        pg = (amg.data_ptr(), bmg.data_ptr(), cmg.data_ptr(), mdg.data_ptr())
    else:
        pc = pg = (None, None, None, None)
    assert FK.jlm_gemm_nt(A.data_ptr(), lda, pc[0], B.data_ptr(), ldb, pc[1], C.data_ptr(), ldc, pc[2], bias.data_ptr(),
                          M, N, K, pc[3], 0) == 0
    assert L.jlm_gemm_nt(Ag.data_ptr(), lda, pg[0], Bg.data_ptr(), ldb, pg[1], Cg.data_ptr(), ldc, pg[2],
                         biasg.data_ptr(), M, N, K, pg[3], _st()) == 0
    torch.cuda.synchronize()
    got, want = Cg.cpu().numpy(), C.numpy()
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert (got == 7.0).sum() == (want == 7.0).sum()        # untouched cells stay untouched


@pytest.mark.parametrize("H,E,R,use_rows", [(64, 32, 10, False), (64, 32, 200, True), (512, 256, 700, True),
                                            (512, 200, 2560, True), (128, 352, 33, True)])
def test_lstm_step(L, H, E, R, use_rows):
    rng = np.random.default_rng(H + E + R)
    V, G = 500, R * 3 + 7
    kpad = (H + E + 31) // 32 * 32
    h, hg = _pair(rng.standard_normal((G, H)).astype(np.float32) * 0.5)
    c, cg = _pair(rng.standard_normal((G, H)).astype(np.float32) * 0.5)
    emb, embg = _pair(rng.standard_normal((V, E)).astype(np.float32) * 0.3)
    wt_np = np.zeros((4 * H, kpad), dtype=np.float32)
    wt_np[:, :H + E] = rng.standard_normal((4 * H, H + E)).astype(np.float32) * 0.08
    wt, wtg = _pair(wt_np)
    bias, biasg = _pair(rng.standard_normal(4 * H).astype(np.float32) * 0.1)
    word, wordg = _pair(rng.integers(0, V, size=G).astype(np.int32))
    if use_rows:
        rows_np = (G - 1 - rng.permutation(R)).astype(np.int32)            # rows to write: top of the store
        prev_np = rng.integers(-1, G - R, size=G).astype(np.int32)         # states read: below them (or zero state)
        rows, rowsg = _pair(rows_np)
        nd, ndg = _pair(np.array([R - 1], dtype=np.int32))
        rp, rpg, ndp, ndpg = rows.data_ptr(), rowsg.data_ptr(), nd.data_ptr(), ndg.data_ptr()
        ho, hog, co, cog = h, hg, c, cg                                    # in place, like the decoder
    else:
        prev_np = np.arange(G, dtype=np.int32)
        rp = rpg = ndp = ndpg = None
        ho, hog = _pair(np.zeros((G, H), dtype=np.float32))
        co, cog = _pair(np.zeros((G, H), dtype=np.float32))
    prev, prevg = _pair(prev_np)
    assert FK.jlm_lstm_step(h.data_ptr(), c.data_ptr(), H, ho.data_ptr(), co.data_ptr(), rp, prev.data_ptr(),
                            word.data_ptr(), emb.data_ptr(), E, wt.data_ptr(), bias.data_ptr(), kpad, H, E, R, ndp, 0) == 0
    assert L.jlm_lstm_step(hg.data_ptr(), cg.data_ptr(), H, hog.data_ptr(), cog.data_ptr(), rpg, prevg.data_ptr(),
                           wordg.data_ptr(), embg.data_ptr(), E, wtg.data_ptr(), biasg.data_ptr(), kpad, H, E, R,
                           ndpg, _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(hog.cpu().numpy(), ho.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cog.cpu().numpy(), co.numpy(), rtol=2e-5, atol=2e-6)


def _unsplit(t):
    """split rows (torch f32-typed storage) -> float64 values [rows, ld]"""
    a = t.detach().cpu().contiguous().numpy().view(np.float16)
    rows, ld2 = a.shape
    sp = a.reshape(rows, ld2 // 16, 2, 8).astype(np.float64)
    return (sp[:, :, 0, :] + sp[:, :, 1, :]).reshape(rows, ld2 // 2)


def _pack(L, src_g, k, scale, ld_dst=None, dst=None, col0=0, src_col0=0):
    """jlm_pack_split_f16 of columns [src_col0, src_col0 + k) of a gpu matrix into dst columns col0.."""
    rows, ld = src_g.shape
    k16 = (k + 15) // 16 * 16
    if dst is None:
        dst = torch.zeros((rows, ld_dst or k16), dtype=torch.float32, device="cuda")
    assert L.jlm_pack_split_f16(src_g.data_ptr() + 4 * src_col0, rows, k, ld, float(scale), dst.data_ptr() + 4 * col0,
                                dst.shape[1], _st()) == 0
    return dst


@pytest.mark.parametrize("chunk", ["48", "256"])
def test_beam_step_chunked_on_ordinary_cells(chunk):
    """JLM_BEAM_CHUNK: every beam step through the chunked kernel (csrc/jlm_beam.hip beam_step_chunked_kernel) with chunks of 48 / 256
    candidates, so that ordinary cells are cut into several -- the kernel tests of all three modes incl. the fused fold of the vocabulary
    partials, and the decodes whose per-frame beams the reference's traces pin (static, vocabulary selection, incremental).  The
    variable is read once per process, hence the child."""
    import os, subprocess, sys
    env = dict(os.environ, JLM_BEAM_CHUNK=chunk)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), os.path.join(here, "test_gpu_decode.py"), "-q", "-x", "-m", "gpu",
                        "-k", "(test_beam_step_and_backtrace or test_beam_step_fused_combine or test_per_frame_beams_match_reference_traces)"
                              " and not 64-6-300"],        # (19 k candidates in chunks of 48: more chunk winners than LDS)
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]


@pytest.mark.parametrize("variant", ["2", "3", "4"])
def test_lstm_step_xg_forced_forms(variant):
    """Every H = 512 form of the LSTM step on every row count of test_lstm_step_xg, whatever the launcher would pick by the row bound:
    JLM_GATE_V = 2 (W-stationary), 3 (persistent 160 x 128), 4 (persistent 128 x 256, a 2 x 2 register block per wave: csrc/jlm_gate_p2.hip;
    its tiles: first / middle / last of a sequence at 5 200 and 20 480 rows, a ragged last tile, a single row).  The variable is read once
    per process, hence the child."""
    import os, subprocess, sys
    env = dict(os.environ, JLM_GATE_V=variant)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_lstm_step_xg and not forced"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]


@pytest.mark.parametrize("H,R,use_rows", [(64, 10, False), (64, 200, True), (512, 700, True), (512, 2560, True), (128, 161, True),
                                         (512, 159, False),
                                         # the W-stationary persistent kernel (H = 512 with a row list): one tile per workgroup at 2 560 rows; two
                                         # or three tiles per workgroup (first / middle / last tile of a sequence); eight (BASELINE configs[2])
                                         (512, 5200, True), (512, 20480, True), (512, 1, True)])
def test_lstm_step_xg(L, H, R, use_rows):
    """jlm_lstm_step_xg (one 160 x 128 tile per CU, gate-interleave-8 order, table rows as accumulator start values)
    against the f64 restatement on the ORIGINAL f32 operands and against the numpy double fed the same split rows;
    in-place state arrays, gathered rows, a device-side row count one short of the bound"""
    rng = np.random.default_rng(7 * H + R)
    V, G = 300, R * 3 + 7
    h_np = np.tanh(rng.standard_normal((G, H))).astype(np.float32)
    c_np = (rng.standard_normal((G, H)) * 0.5).astype(np.float32)
    W = (rng.standard_normal((4 * H, H)) * 0.08).astype(np.float32)             # rows in gate-interleave-8 order
    xg = (rng.standard_normal((V, 4 * H)) * 0.6).astype(np.float32)             # emb . W_x^T + b, same order
    word_np = rng.integers(0, V, size=G).astype(np.int32)
    S = 20
    if use_rows:
        rows_np = (G - 1 - rng.permutation(R)).astype(np.int32)
        prev_np = rng.integers(-1, G - R, size=G).astype(np.int32)
        n_live = R - 1
    else:
        rows_np = None
        prev_np = np.concatenate([np.full(R, -1), np.arange(G - R)]).astype(np.int32)   # rows 0..R-1 <- zero state / R.. <- earlier
        prev_np = np.where(np.arange(G) < R, np.where(np.arange(G) % 3 == 0, -1, R + np.arange(G) % (G - R)), 0).astype(np.int32)
        n_live = R
    sel = rows_np[:n_live] if use_rows else np.arange(R)
    # f64 restatement
    p = prev_np[sel].astype(np.int64)
    hin = np.where((p >= 0)[:, None], h_np[np.maximum(p, 0)], 0.0).astype(np.float64)
    cin = np.where((p >= 0)[:, None], c_np[np.maximum(p, 0)], 0.0).astype(np.float64)
    z = hin @ W.astype(np.float64).T + xg[word_np[sel]].astype(np.float64)
    u = np.arange(H)
    zi, zf, zo, zg = (z[:, (u // 8) * 32 + k * 8 + (u % 8)] for k in range(4))
    sig = lambda t: 1.0 / (np.exp(-t) + 1.0)
    cn = cin * sig(zf) + np.tanh(zg) * sig(zi)
    hn = np.tanh(cn) * sig(zo)
    # device operands
    hg, cg = torch.as_tensor(h_np).cuda(), torch.as_tensor(c_np.copy()).cuda()
    hs = _pack(L, hg, H, 2.0 ** 14)
    ws = _pack(L, torch.as_tensor(W).cuda(), H, 2.0 ** (S - 14))
    xg8 = torch.as_tensor(xg * np.float32(2.0 ** S)).cuda()
    wordg, prevg = torch.as_tensor(word_np).cuda(), torch.as_tensor(prev_np).cuda()
    rowsg = torch.as_tensor(rows_np).cuda() if use_rows else None
    ndg = torch.as_tensor(np.array([n_live], dtype=np.int32)).cuda() if use_rows else None
    torch.cuda.synchronize()
    hs_c, cc = hs.cpu(), torch.as_tensor(c_np.copy())
    hf32 = torch.full((G, H), 7.0, dtype=torch.float32, device="cuda")      # optional plain f32 copy of h'
    args = (H, 2.0 ** -S, 2.0 ** 14, hf32.data_ptr(), R)
    assert L.jlm_lstm_step_xg(hs.data_ptr(), cg.data_ptr(), H, hs.data_ptr(), cg.data_ptr(), rowsg.data_ptr() if use_rows else None,
                              prevg.data_ptr(), wordg.data_ptr(), ws.data_ptr(), xg8.data_ptr(), *args,
                              ndg.data_ptr() if use_rows else None, _st()) == 0
    torch.cuda.synchronize()
    h_gpu = _unsplit(hs) / 2.0 ** 14
    c_gpu = cg.cpu().numpy()
    np.testing.assert_allclose(h_gpu[sel], hn, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(c_gpu[sel], cn, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(hf32.cpu().numpy()[sel], hn, rtol=2e-5, atol=2e-6)      # the f32 copy = the split rows' value
    # rows that were not stepped are untouched
    mask = np.ones(G, dtype=bool)
    mask[sel] = False
    np.testing.assert_array_equal(c_gpu[mask], c_np[mask])
    assert (hf32.cpu().numpy()[mask] == 7.0).all()
    # the numpy double on the same split rows
    ws_c, xg_c = ws.cpu(), xg8.cpu()
    rows_c = torch.as_tensor(rows_np) if use_rows else None
    nd_c = torch.as_tensor(np.array([n_live], dtype=np.int32)) if use_rows else None
    word_c, prev_c = torch.as_tensor(word_np), torch.as_tensor(prev_np)
    hf32_c = torch.zeros((G, H), dtype=torch.float32)
    assert FK.jlm_lstm_step_xg(hs_c.data_ptr(), cc.data_ptr(), H, hs_c.data_ptr(), cc.data_ptr(),
                               rows_c.data_ptr() if use_rows else None, prev_c.data_ptr(), word_c.data_ptr(), ws_c.data_ptr(),
                               xg_c.data_ptr(), H, 2.0 ** -S, 2.0 ** 14, hf32_c.data_ptr(), R,
                               nd_c.data_ptr() if use_rows else None, 0) == 0
    np.testing.assert_allclose(h_gpu[sel], (_unsplit(hs_c) / 2.0 ** 14)[sel], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(c_gpu[sel], cc.numpy()[sel], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("V,K,R", [(300, 64, 40), (5000, 512, 700), (1237, 128, 2560)])
def test_vocab_lse_partials_split(L, V, K, R):
    """tile-form normaliser on split rows (untied models: k = H > 256): per 128-word tile (max, sum exp) of
    descale * B.T + bias against the f64 evaluation of the original f32 operands"""
    rng = np.random.default_rng(V + K + R)
    G = R + 50
    B_np = (rng.standard_normal((V, K)) * 0.08).astype(np.float32)
    T_np = np.tanh(rng.standard_normal((G, K))).astype(np.float32)
    bias_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
    rows_np = rng.permutation(G)[:R].astype(np.int32)
    Bs = _pack(L, torch.as_tensor(B_np).cuda(), K, 2.0 ** 10)
    Ts = _pack(L, torch.as_tensor(T_np).cuda(), K, 2.0 ** 14)
    bias, rows = torch.as_tensor(bias_np).cuda(), torch.as_tensor(rows_np).cuda()
    nd = torch.as_tensor(np.array([R - 3], dtype=np.int32)).cuda()
    ntile = (V + 127) // 128
    part = torch.zeros((ntile + 2, R, 2), dtype=torch.float32, device="cuda")
    r = L.jlm_vocab_lse_partials_split(Bs.data_ptr(), K, V, K, Ts.data_ptr(), K, rows.data_ptr(), bias.data_ptr(), 2.0 ** -24,
                                       part.data_ptr(), R, 2, R, nd.data_ptr(), _st())
    assert r == ntile
    torch.cuda.synchronize()
    p = part.cpu().numpy().astype(np.float64)
    y = T_np[rows_np[:R - 3]].astype(np.float64) @ B_np.astype(np.float64).T + bias_np
    for t in range(ntile):
        yt = y[:, t * 128:(t + 1) * 128]
        want = np.log(np.exp(yt - yt.max(axis=1, keepdims=True)).sum(axis=1)) + yt.max(axis=1)
        got = p[2 + t, :R - 3, 0] + np.log(p[2 + t, :R - 3, 1])
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    # the numpy double of the same contract
    part_c = torch.zeros((ntile + 2, R, 2), dtype=torch.float32)
    assert FK.jlm_vocab_lse_partials_split(Bs.cpu().data_ptr(), K, V, K, Ts.cpu().data_ptr(), K, rows.cpu().data_ptr(),
                                           bias.cpu().data_ptr(), 2.0 ** -24, part_c.data_ptr(), R, 2, R, nd.cpu().data_ptr(), 0) == ntile


@pytest.mark.parametrize("M,N,K,maps", [(70, 40, 32, False), (300, 352, 512, True), (2560, 352, 512, True), (1, 8, 16, False)])
def test_gemm_nt_split(L, M, N, K, maps):
    rng = np.random.default_rng(M + N + K)
    GA = M + 9
    A, Ag = _pair(rng.standard_normal((GA, K)).astype(np.float32))
    Bm, Bg = _pair(rng.standard_normal((N, K)).astype(np.float32) * 0.1)
    bias, biasg = _pair(rng.standard_normal(N).astype(np.float32))
    ldc = N + 4
    C, Cg = _pair(np.zeros((GA, ldc), dtype=np.float32))
    if maps:
        ar, arg = _pair(rng.permutation(GA)[:M].astype(np.int32))
        nd, ndg = _pair(np.array([M - 1], dtype=np.int32))
        arp, argp, ndp, ndgp = ar.data_ptr(), arg.data_ptr(), nd.data_ptr(), ndg.data_ptr()
    else:
        arp = argp = ndp = ndgp = None
    assert FK.jlm_gemm_nt(A.data_ptr(), K, arp, Bm.data_ptr(), K, None, C.data_ptr(), ldc, arp, bias.data_ptr(), M, N, K,
                          ndp, 0) == 0
    As, Bs = _pack(L, Ag, K, 2.0 ** 10), _pack(L, Bg, K, 2.0 ** 12)
    assert L.jlm_gemm_nt_split(As.data_ptr(), K, argp, Bs.data_ptr(), K, None, Cg.data_ptr(), ldc, argp, biasg.data_ptr(),
                               2.0 ** -22, M, N, K, ndgp, _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(Cg.cpu().numpy(), C.numpy(), rtol=2e-5, atol=1e-5)
    torch.cuda.synchronize()
    C2 = torch.zeros((GA, ldc), dtype=torch.float32)
    As_c, Bs_c = As.cpu(), Bs.cpu()
    assert FK.jlm_gemm_nt_split(As_c.data_ptr(), K, arp, Bs_c.data_ptr(), K, None, C2.data_ptr(), ldc, arp, bias.data_ptr(),
                                2.0 ** -22, M, N, K, ndp, 0) == 0
    np.testing.assert_allclose(Cg.cpu().numpy(), C2.numpy(), rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("V,K,R", [(1000, 32, 10), (50000, 256, 300), (12000, 200, 2560), (20000, 52, 777), (130, 100, 129)])
def test_vocab_lse(L, V, K, R):
    rng = np.random.default_rng(V + K + R)
    ldt, G = K + 12, R + 40
    Bm, Bg = _pair(rng.standard_normal((V, K)).astype(np.float32) * 0.3)
    T, Tg = _pair(rng.standard_normal((G, ldt)).astype(np.float32))
    bias, biasg = _pair(rng.standard_normal(V).astype(np.float32))
    rows, rowsg = _pair(rng.permutation(G)[:R].astype(np.int32))
    nd, ndg = _pair(np.array([R - 2], dtype=np.int32))
    ntile = (V + 127) // 128
    part, partg = _pair(np.zeros((ntile + 3, R, 2), dtype=np.float32))
    lse, lseg = _pair(np.zeros(G, dtype=np.float64))
    t_off = 8
    r0 = FK.jlm_vocab_lse_partials(Bm.data_ptr(), K, V, K, T.data_ptr() + 4 * t_off, ldt, rows.data_ptr(), bias.data_ptr(),
                                   part.data_ptr(), R, 3, R, nd.data_ptr(), 0)
    r1 = L.jlm_vocab_lse_partials(Bg.data_ptr(), K, V, K, Tg.data_ptr() + 4 * t_off, ldt, rowsg.data_ptr(),
                                  biasg.data_ptr(), partg.data_ptr(), R, 3, R, ndg.data_ptr(), _st())
    assert r0 == r1 == ntile
    # tile 0..2 left for "another segment": fill with neutral partials
    for p in (part, partg):
        p[:3, :, 0] = -3.0e38
        p[:3, :, 1] = 0.0
    assert FK.jlm_lse_combine(part.data_ptr(), R, ntile + 3, rows.data_ptr(), lse.data_ptr(), R, nd.data_ptr(), 0) == 0
    assert L.jlm_lse_combine(partg.data_ptr(), R, ntile + 3, rowsg.data_ptr(), lseg.data_ptr(), R, ndg.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    n = R - 2
    pg, pc = partg.cpu().numpy()[3:, :n], part.numpy()[3:, :n]
    np.testing.assert_allclose(pg[..., 0], pc[..., 0], rtol=1e-5, atol=2e-5)
    lse_tile_g = pg[..., 0] + np.log(pg[..., 1])
    lse_tile_c = pc[..., 0] + np.log(pc[..., 1])
    np.testing.assert_allclose(lse_tile_g, lse_tile_c, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(lseg.cpu().numpy(), lse.numpy(), rtol=1e-6, atol=2e-5)


@pytest.mark.parametrize("V,widths,R", [(2000, [32], 10), (3000, [32, 16, 8], 300), (50000, [200, 100, 52], 2560),
                                        (50000, [256], 1000), (700, [100], 129), (5000, [8, 256, 40], 64)])
def test_vocab_lse_stationary(L, V, widths, R):
    rng = np.random.default_rng(V + R + len(widths))
    segs_c, segs_g, keep, ldt = _segments(rng, V, widths, ldt_extra=8)
    G = R + 50
    T, Tg = _pair(rng.standard_normal((G, ldt)).astype(np.float32))
    b2, b2g = _pair(rng.standard_normal(V).astype(np.float32))
    rows, rowsg = _pair(rng.permutation(G)[:R].astype(np.int32))
    nd, ndg = _pair(np.array([R - 1], dtype=np.int32))
    maxp = 96
    part, partg = _pair(np.zeros((maxp, R, 2), dtype=np.float32))
    lse, lseg = _pair(np.zeros(G, dtype=np.float64))
    n0 = FK.jlm_vocab_lse_stationary(segs_c, len(widths), b2.data_ptr(), T.data_ptr(), ldt, rows.data_ptr(), part.data_ptr(),
                                     R, maxp, R, nd.data_ptr(), 0)
    n1 = L.jlm_vocab_lse_stationary(segs_g, len(widths), b2g.data_ptr(), Tg.data_ptr(), ldt, rowsg.data_ptr(),
                                    partg.data_ptr(), R, maxp, R, ndg.data_ptr(), _st())
    assert n0 == len(widths) and len(widths) <= n1 <= maxp, (n0, n1)
    assert FK.jlm_lse_combine(part.data_ptr(), R, n0, rows.data_ptr(), lse.data_ptr(), R, nd.data_ptr(), 0) == 0
    assert L.jlm_lse_combine(partg.data_ptr(), R, n1, rowsg.data_ptr(), lseg.data_ptr(), R, ndg.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(lseg.cpu().numpy(), lse.numpy(), rtol=1e-6, atol=3e-5)
    # a slice budget smaller than the segment count is an error, k > 256 asks for the tile form
    assert L.jlm_vocab_lse_stationary(segs_g, len(widths), b2g.data_ptr(), Tg.data_ptr(), ldt, rowsg.data_ptr(),
                                      partg.data_ptr(), R, 0, R, ndg.data_ptr(), _st()) < 0


def _split_segments(L, segs_g, keep_g, n, scale_exp, b2g=None):
    """split-row copies of the gpu segments' matrices -> (Segment array, t_scale, descale, bias_col, keepalive);
    with b2g the bias goes into the first padded column wherever a segment has one"""
    import ctypes
    out = (_lib.Segment * n)()
    ts, ds, bc, keep = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_int * n)(), []
    for i in range(n):
        sg = segs_g[i]
        nv, kp = sg.v_end - sg.v_start, (sg.k + 15) // 16 * 16
        dst = torch.zeros((nv, kp), dtype=torch.float32, device="cuda")
        assert L.jlm_pack_split_f16(sg.B, nv, sg.k, sg.ldb, float(2.0 ** scale_exp[i]), dst.data_ptr(), kp, _st()) == 0
        bc[i] = -1
        if b2g is not None and sg.k % 16:
            assert L.jlm_pack_split_f16_col(b2g.data_ptr() + 4 * sg.v_start, nv, float(2.0 ** scale_exp[i]), dst.data_ptr(), kp,
                                            sg.k, _st()) == 0
            bc[i] = sg.k
        keep.append(dst)
        out[i] = _lib.Segment(sg.v_start, sg.v_end, sg.k, sg.t_off, dst.data_ptr(), kp)
        ts[i] = 2.0 ** 3
        ds[i] = 2.0 ** -(3 + scale_exp[i])
    return out, ts, ds, bc, keep


@pytest.mark.parametrize("V,widths,R", [(2000, [32], 10), (3000, [32, 16, 8], 300), (50000, [200, 100, 52], 2560),
                                        (50000, [256], 1000), (700, [100], 129), (5000, [8, 256, 40], 64),
                                        (4000, [160, 112, 64], 200)])
def test_vocab_lse_split(L, V, widths, R):
    """split-f16 form of the rows-stationary LSE: same contract; its error against an f64
    evaluation of the same f32 operands must be of the size of the plain f32 kernel's"""
    rng = np.random.default_rng(V + R + len(widths))
    segs_c, segs_g, keep, ldt = _segments(rng, V, widths, ldt_extra=8)
    G = R + 50
    T, Tg = _pair(rng.standard_normal((G, ldt)).astype(np.float32))
    b2, b2g = _pair(rng.standard_normal(V).astype(np.float32))
    rows, rowsg = _pair(rng.permutation(G)[:R].astype(np.int32))
    nd, ndg = _pair(np.array([R - 1], dtype=np.int32))
    maxp = 96
    partg = torch.zeros((maxp, R, 2), dtype=torch.float32, device="cuda")
    lse64 = np.zeros(G)
    n = R - 1
    ys = []
    for i in range(len(widths)):                 # f64 reference straight from the f32 operands
        sg = segs_c[i]
        nv = sg.v_end - sg.v_start
        Bv = FK_view(sg.B, nv * sg.ldb).reshape(nv, sg.ldb)[:, :sg.k].astype(np.float64)
        Tv = T.numpy()[rows.numpy()[:n], sg.t_off:sg.t_off + sg.k].astype(np.float64)
        ys.append(Tv @ Bv.T + b2.numpy()[sg.v_start:sg.v_end].astype(np.float64))
    y = np.concatenate(ys, axis=1)
    mx = y.max(axis=1)
    ref = mx + np.log(np.exp(y - mx[:, None]).sum(axis=1))
    errs = {}
    for name in ("f32", "f16x3", "f16x3-bias-col"):
        lseg = torch.zeros(G, dtype=torch.float64, device="cuda")
        if name == "f32":
            n1 = L.jlm_vocab_lse_stationary(segs_g, len(widths), b2g.data_ptr(), Tg.data_ptr(), ldt, rowsg.data_ptr(),
                                            partg.data_ptr(), R, maxp, R, ndg.data_ptr(), _st())
        else:
            sp, ts, ds, bc, keep2 = _split_segments(L, segs_g, keep, len(widths), [6] * len(widths),
                                                    b2g if name.endswith("col") else None)
            n1 = L.jlm_vocab_lse_split(sp, ts, ds, bc, len(widths), b2g.data_ptr(), Tg.data_ptr(), ldt, rowsg.data_ptr(),
                                       partg.data_ptr(), R, maxp, R, ndg.data_ptr(), _st())
        assert len(widths) <= n1 <= maxp, n1
        assert L.jlm_lse_combine(partg.data_ptr(), R, n1, rowsg.data_ptr(), lseg.data_ptr(), R, ndg.data_ptr(), _st()) == 0
        torch.cuda.synchronize()
        errs[name] = np.abs(lseg.cpu().numpy()[rows.numpy()[:n]] - ref).max()
    print("max |lse - f64|: f32 MFMA %.3g, split f16 %.3g, with the bias as a GEMM column %.3g" % (
        errs["f32"], errs["f16x3"], errs["f16x3-bias-col"]))
    for name in ("f16x3", "f16x3-bias-col"):
        assert errs[name] < 2e-5
        assert errs[name] < 4 * errs["f32"] + 2e-6


def FK_view(ptr, count):
    from tests.fake_hip import view
    return view(ptr, count, np.float32)


def _segments(rng, V, widths, ldt_extra=0):
    """random multi-segment output side: returns (cpu Segment array, gpu Segment array, keepalive, ldt)"""
    bounds = np.linspace(0, V, len(widths) + 1).astype(int)
    segs_c = (_lib.Segment * len(widths))()
    segs_g = (_lib.Segment * len(widths))()
    keep, off = [], 0
    for i, k in enumerate(widths):
        kp = (k + 3) // 4 * 4
        blk = np.zeros((bounds[i + 1] - bounds[i], kp), dtype=np.float32)
        blk[:, :k] = rng.standard_normal((blk.shape[0], k)).astype(np.float32) * 0.3
        bc, bg = _pair(blk)
        keep += [bc, bg]
        segs_c[i] = _lib.Segment(int(bounds[i]), int(bounds[i + 1]), kp, off, bc.data_ptr(), kp)
        segs_g[i] = _lib.Segment(int(bounds[i]), int(bounds[i + 1]), kp, off, bg.data_ptr(), kp)
        off += kp
    return segs_c, segs_g, keep, off + ldt_extra


def _wordlist_problem(rng, V, widths, beam, n_groups, max_words, dup=False, share_lists=True):
    segs_c, segs_g, keep, ldt = _segments(rng, V, widths)
    G = n_groups * beam + 5
    T = rng.standard_normal((G, ldt)).astype(np.float32)
    b2 = rng.standard_normal(V).astype(np.float32)
    cnt = rng.integers(0, beam + 1, size=n_groups).astype(np.int32)
    cnt[0] = beam
    g0 = (np.arange(n_groups) * beam).astype(np.int32)
    lists = []
    for j in range(n_groups + 3):
        n = int(rng.integers(0, max_words + 1))
        w = rng.integers(0, V, size=n)
        if dup and n > 2:
            w[1] = w[0]
        lists.append(w.astype(np.int32))
    off = np.zeros(len(lists) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(x) for x in lists])
    wl = np.concatenate(lists + [np.zeros(1, np.int32)])
    cidx = rng.permutation(n_groups).astype(np.int32)
    cnt_store = np.zeros(n_groups, dtype=np.int32)
    cnt_store[cidx] = cnt
    if share_lists:     # several groups may read the same list (as the incremental decoder's deltas do)
        wl_idx = rng.integers(0, 3, size=n_groups).astype(np.int32) + np.arange(n_groups, dtype=np.int32) - 2
    else:               # edge logits: one list per group (every edge is written once)
        wl_idx = rng.permutation(n_groups + 1)[:n_groups].astype(np.int32) - 2
    return dict(segs_c=segs_c, segs_g=segs_g, keep=keep, ldt=ldt, T=T, b2=b2, cnt=cnt_store, cidx=cidx, g0=g0, wl=wl,
                off=off, wl_idx=wl_idx, G=G, n_words=int(off[-1]))


@pytest.mark.parametrize("widths,beam,ng,mw", [([32], 10, 20, 40), ([200, 100, 50], 10, 64, 90), ([256], 20, 9, 700),
                                               ([32, 16, 8], 3, 30, 5)])
def test_edge_logits(L, widths, beam, ng, mw):
    rng = np.random.default_rng(sum(widths) + beam + ng)
    P = _wordlist_problem(rng, 3000, widths, beam, ng, mw, share_lists=False)
    nseg = len(widths)
    out_ids = rng.permutation(P["n_words"] + 7).astype(np.int32)[:P["n_words"] + 1]
    ten = {k: _pair(P[k]) for k in ("T", "b2", "cnt", "cidx", "g0", "wl", "off", "wl_idx")}
    oi, oig = _pair(out_ids)
    edge, edgeg = _pair(np.full((P["n_words"] + 8) * beam, 5.0, dtype=np.float32))
    a = lambda k, i: ten[k][i].data_ptr()
    assert FK.jlm_edge_logits(P["segs_c"], nseg, a("b2", 0), a("T", 0), P["ldt"], a("g0", 0), a("cnt", 0), a("cidx", 0),
                              a("wl", 0), a("off", 0), a("wl_idx", 0), 2, oi.data_ptr(), edge.data_ptr(), beam, ng, 0) == 0
    assert L.jlm_edge_logits(P["segs_g"], nseg, a("b2", 1), a("T", 1), P["ldt"], a("g0", 1), a("cnt", 1), a("cidx", 1),
                             a("wl", 1), a("off", 1), a("wl_idx", 1), 2, oig.data_ptr(), edgeg.data_ptr(), beam, ng,
                             _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(edgeg.cpu().numpy(), edge.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("widths,beam,ng,mw,merge", [([32], 10, 20, 40, 0), ([200, 100, 50], 10, 64, 90, 1),
                                                     ([256], 20, 9, 700, 0), ([256], 20, 9, 700, 1),
                                                     ([32, 16, 8], 3, 30, 5, 1), ([256], 48, 7, 300, 1), ([128], 64, 5, 200, 0)])
def test_wordlist_lse(L, widths, beam, ng, mw, merge):
    rng = np.random.default_rng(sum(widths) + beam + ng + merge)
    P = _wordlist_problem(rng, 3000, widths, beam, ng, mw, dup=True)
    nseg = len(widths)
    ten = {k: _pair(P[k]) for k in ("T", "b2", "cnt", "cidx", "g0", "wl", "off", "wl_idx")}
    rm, rmg = _pair(rng.standard_normal(P["G"]).astype(np.float32))
    rs, rsg = _pair(rng.uniform(1.0, 50.0, size=P["G"]))
    ls, lsg = _pair(np.full(P["G"], 123.0))
    a = lambda k, i: ten[k][i].data_ptr()
    assert FK.jlm_wordlist_lse(P["segs_c"], nseg, a("b2", 0), a("T", 0), P["ldt"], a("g0", 0), a("cnt", 0), a("cidx", 0),
                               a("wl", 0), a("off", 0), a("wl_idx", 0), 2, rm.data_ptr(), rs.data_ptr(), ls.data_ptr(),
                               merge, beam, ng, 0) == 0
    assert L.jlm_wordlist_lse(P["segs_g"], nseg, a("b2", 1), a("T", 1), P["ldt"], a("g0", 1), a("cnt", 1), a("cidx", 1),
                              a("wl", 1), a("off", 1), a("wl_idx", 1), 2, rmg.data_ptr(), rsg.data_ptr(), lsg.data_ptr(),
                              merge, beam, ng, _st()) == 0
    torch.cuda.synchronize()
    got, want = lsg.cpu().numpy(), ls.numpy()
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all()
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6, atol=3e-5)
    lg = rmg.cpu().numpy().astype(np.float64) + np.log(rsg.cpu().numpy())
    lc = rm.numpy().astype(np.float64) + np.log(rs.numpy())
    np.testing.assert_allclose(lg[fin], lc[fin], rtol=1e-6, atol=3e-5)


@pytest.mark.parametrize("width,beam,ng,mw,merge", [(32, 10, 20, 40, 0), (256, 20, 9, 700, 0), (256, 20, 9, 700, 1),
                                                    (100, 10, 64, 300, 1), (200, 32, 5, 2500, 0), (48, 1, 7, 33, 0),
                                                    (256, 48, 6, 400, 1), (100, 64, 4, 300, 0), (64, 33, 9, 150, 1)])
def test_wordlist_lse_split(L, width, beam, ng, mw, merge):
    """split-f16 word-list LSE (deep gather ring) against the f64 restatement on the original f32 operands"""
    rng = np.random.default_rng(width + beam + ng + merge)
    P = _wordlist_problem(rng, 3000, [width], beam, ng, mw, dup=True)
    ten = {k: _pair(P[k]) for k in ("T", "b2", "cnt", "cidx", "g0", "wl", "off", "wl_idx")}
    rm, rmg = _pair(rng.standard_normal(P["G"]).astype(np.float32))
    rs, rsg = _pair(rng.uniform(1.0, 50.0, size=P["G"]))
    ls, lsg = _pair(np.full(P["G"], 123.0))
    a = lambda k, i: ten[k][i].data_ptr()
    assert FK.jlm_wordlist_lse(P["segs_c"], 1, a("b2", 0), a("T", 0), P["ldt"], a("g0", 0), a("cnt", 0), a("cidx", 0),
                               a("wl", 0), a("off", 0), a("wl_idx", 0), 2, rm.data_ptr(), rs.data_ptr(), ls.data_ptr(),
                               merge, beam, ng, 0) == 0
    sp, ts, ds, bc, keep = _split_segments(L, P["segs_g"], None, 1, [6])
    max_words = int(np.diff(P["off"]).max())
    assert L.jlm_wordlist_lse_split(sp, ts[0], ds[0], a("b2", 1), a("T", 1), P["ldt"], a("g0", 1), a("cnt", 1), a("cidx", 1),
                                    a("wl", 1), a("off", 1), a("wl_idx", 1), 2, max_words, rmg.data_ptr(), rsg.data_ptr(),
                                    lsg.data_ptr(), merge, beam, ng, _st()) == 0
    torch.cuda.synchronize()
    got, want = lsg.cpu().numpy(), ls.numpy()
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all()
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6, atol=3e-5)
    lg = rmg.cpu().numpy().astype(np.float64) + np.log(rsg.cpu().numpy())
    lc = rm.numpy().astype(np.float64) + np.log(rs.numpy())
    np.testing.assert_allclose(lg[fin], lc[fin], rtol=1e-6, atol=3e-5)


@pytest.mark.parametrize("width,beam,B,nf,mw", [(256, 10, 7, 6, 40), (100, 20, 3, 19, 128), (32, 3, 20, 2, 5), (200, 10, 64, 12, 70),
                                                (256, 48, 5, 4, 60), (100, 64, 3, 5, 100)])
def test_wordlist_merge_split(L, width, beam, B, nf, mw):
    """per-sentence merge of a frame's new words into all older rows against its numpy restatement"""
    rng = np.random.default_rng(width + beam + B + nf)
    V = 3000
    segs_c, segs_g, keep, ldt = _segments(rng, V, [width])
    rmax, G = B * beam, nf * B * beam
    T, Tg = _pair(rng.standard_normal((G, ldt)).astype(np.float32))
    b2, b2g = _pair(rng.standard_normal(V).astype(np.float32))
    cnt, cntg = _pair(rng.integers(0, beam + 1, size=nf * B).astype(np.int32))
    lists = [rng.integers(0, V, size=int(rng.integers(0, mw + 1))).astype(np.int32) for _ in range(3 * B)]
    lists[2 * B] = rng.integers(0, V, size=mw).astype(np.int32)
    off = np.zeros(len(lists) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(x) for x in lists])
    wl, wlg = _pair(np.concatenate(lists + [np.zeros(1, np.int32)]))
    offt, offg = _pair(off)
    rm, rmg = _pair(rng.standard_normal(G).astype(np.float32))
    rs, rsg = _pair(rng.uniform(1.0, 50.0, size=G))
    ls, lsg = _pair(np.full(G, 123.0))
    sp, ts, ds, bc, keep2 = _split_segments(L, segs_g, keep, 1, [6])
    sp_c = (_lib.Segment * 1)()
    host_rows = keep2[0].cpu()
    sp_c[0] = _lib.Segment(sp[0].v_start, sp[0].v_end, sp[0].k, sp[0].t_off, host_rows.data_ptr(), sp[0].ldb)
    args = lambda x: (ts[0], ds[0])
    assert FK.jlm_wordlist_merge_split(sp_c, ts[0], ds[0], b2.data_ptr(), T.data_ptr(), ldt, cnt.data_ptr(), B, beam, nf,
                                       wl.data_ptr(), offt.data_ptr(), 2 * B, mw, rm.data_ptr(), rs.data_ptr(),
                                       ls.data_ptr(), 0) == 0
    assert L.jlm_wordlist_merge_split(sp, ts[0], ds[0], b2g.data_ptr(), Tg.data_ptr(), ldt, cntg.data_ptr(), B, beam, nf,
                                      wlg.data_ptr(), offg.data_ptr(), 2 * B, mw, rmg.data_ptr(), rsg.data_ptr(),
                                      lsg.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(lsg.cpu().numpy(), ls.numpy(), rtol=1e-6, atol=3e-5)
    lg = rmg.cpu().numpy().astype(np.float64) + np.log(rsg.cpu().numpy())
    lc = rm.numpy().astype(np.float64) + np.log(rs.numpy())
    np.testing.assert_allclose(lg, lc, rtol=1e-6, atol=3e-5)
    assert L.jlm_wordlist_merge_split(sp, ts[0], ds[0], b2g.data_ptr(), Tg.data_ptr(), ldt, cntg.data_ptr(), B, beam, nf,
                                      wlg.data_ptr(), offg.data_ptr(), 2 * B, 129, rmg.data_ptr(), rsg.data_ptr(),
                                      lsg.data_ptr(), _st()) == -2


def _beam_problem(rng, B, beam, F, max_nodes):
    """random lattice + beam state, consistent up to frame F-1"""
    rmax, G = B * beam, F * B * beam
    slen = rng.integers(1, F, size=B).astype(np.int32)
    slen[0] = F - 1
    counts = np.zeros(F * B, dtype=np.int64)
    starts, words = [], []
    for f in range(F):
        for s in range(B):
            if f == 0:
                n = 1
            elif f <= slen[s]:
                n = int(rng.integers(1, max_nodes + 1))
            else:
                n = 0
            counts[f * B + s] = n
            for _ in range(n):
                starts.append(-1 if f == 0 else int(rng.integers(max(0, f - 4), f)))
                words.append(int(rng.integers(0, 1000)))
    end_off = np.zeros(F * B + 1, dtype=np.int32)
    end_off[1:] = np.cumsum(counts)
    N = int(end_off[-1])
    return dict(B=B, beam=beam, F=F, rmax=rmax, G=G, N=N, slen=slen, end_off=end_off,
                nstart=np.array(starts, dtype=np.int32), nword=np.array(words, dtype=np.int32),
                max_cands=int(counts.max()) * beam)


def _run_beam(lib, P, mode, cuda, rng_seed):
    rng = np.random.default_rng(rng_seed)
    dev = "cuda" if cuda else "cpu"
    t = lambda a: torch.as_tensor(a).contiguous().to(dev)
    G, F, B, beam = P["G"], P["F"], P["B"], P["beam"]
    ints = {k: t(P[k]) for k in ("slen", "end_off", "nstart", "nword")}
    score, lse, ysum = t(np.zeros(G)), t(rng.uniform(5.0, 9.0, size=G)), t(np.zeros(G))
    # make some exact ties: quantise the edge logits coarsely
    edge = t((np.round(rng.standard_normal(max(P["N"], 1) * beam) * 2.0) / 2.0).astype(np.float32))
    bp, node, word = (t(np.full(G, -7, dtype=np.int32)) for _ in range(3))
    cnt, live, n_live = t(np.zeros(F * B, dtype=np.int32)), t(np.full(G, -1, dtype=np.int32)), t(np.zeros(F, dtype=np.int32))
    lat = _lib.Lattice(B, beam, F, ints["slen"].data_ptr(), ints["end_off"].data_ptr(), ints["nstart"].data_ptr(),
                       ints["nword"].data_ptr())
    st = _lib.BeamState(score.data_ptr(), lse.data_ptr(), ysum.data_ptr(), bp.data_ptr(), node.data_ptr(), word.data_ptr(),
                        cnt.data_ptr(), live.data_ptr(), n_live.data_ptr(), edge.data_ptr())
    stream = _st() if cuda else 0
    for f in range(F):
        assert lib.jlm_beam_step(lat, st, f, mode, P["max_cands"], stream) == 0
    stride = F + 1
    on, ol, osc = t(np.full((P["rmax"], stride), -1, dtype=np.int32)), t(np.zeros(P["rmax"], dtype=np.int32)), t(np.zeros(P["rmax"]))
    assert lib.jlm_backtrace(lat, st, on.data_ptr(), ol.data_ptr(), osc.data_ptr(), stride, stream) == 0
    if cuda:
        torch.cuda.synchronize()
    out = dict(score=score, ysum=ysum, bp=bp, node=node, word=word, cnt=cnt, live=live, n_live=n_live, on=on, ol=ol, osc=osc)
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,beam,F,max_nodes", [(7, 10, 9, 12), (64, 3, 6, 40), (3, 20, 22, 70),
                                                # cells above what one wave's LDS holds in one piece (13 312 candidates at beam 64): the
                                                # chunked kernel (round 6) -- before it such sentences left the batch for a host-side search
                                                (3, 64, 6, 300)])
def test_beam_step_and_backtrace(L, mode, B, beam, F, max_nodes):
    P = _beam_problem(np.random.default_rng(B * 100 + beam + F), B, beam, F, max_nodes)
    if max_nodes >= 300:
        assert P["max_cands"] > 13312 and L.jlm_beam_step_max_cands(beam, F, mode) > P["max_cands"]
    want = _run_beam(FK, P, mode, False, 5)
    got = _run_beam(L, P, mode, True, 5)
    np.testing.assert_array_equal(got["cnt"], want["cnt"])
    np.testing.assert_array_equal(got["n_live"], want["n_live"])
    rmax = P["rmax"]
    for f in range(F):
        for s in range(B):
            k = int(want["cnt"][f * B + s])
            sl = slice(f * rmax + s * beam, f * rmax + s * beam + k)
            for name in ("bp", "node", "word"):
                np.testing.assert_array_equal(got[name][sl], want[name][sl], err_msg="%s f=%d s=%d" % (name, f, s))
            np.testing.assert_allclose(got["score"][sl], want["score"][sl], rtol=0, atol=1e-12)
            if mode == 2:
                np.testing.assert_allclose(got["ysum"][sl], want["ysum"][sl], rtol=0, atol=1e-12)
        nl = int(want["n_live"][f])
        assert sorted(got["live"][f * rmax:f * rmax + nl]) == sorted(want["live"][f * rmax:f * rmax + nl])
    np.testing.assert_array_equal(got["ol"], want["ol"])
    np.testing.assert_allclose(got["osc"], want["osc"], rtol=0, atol=1e-12)
    for i in range(rmax):
        np.testing.assert_array_equal(got["on"][i, :want["ol"][i]], want["on"][i, :want["ol"][i]])


def _run_beam_fused(lib, P, cuda, n_parts):
    """mode 0 with the log-normalisers arriving as partial (max, sum exp) slices indexed by live
    position, folded inside jlm_beam_step (jlm_beam_state.lse_part)"""
    dev = "cuda" if cuda else "cpu"
    t = lambda a: torch.as_tensor(a).contiguous().to(dev)
    G, F, B, beam, rmax = P["G"], P["F"], P["B"], P["beam"], P["rmax"]
    rng = np.random.default_rng(11)
    ints = {k: t(P[k]) for k in ("slen", "end_off", "nstart", "nword")}
    score, lse = t(np.zeros(G)), t(np.full(G, 1e30))                    # lse must come from the slices
    edge = t((np.round(rng.standard_normal(max(P["N"], 1) * beam) * 2.0) / 2.0).astype(np.float32))
    bp, node, word = (t(np.full(G, -7, dtype=np.int32)) for _ in range(3))
    cnt, live, n_live = t(np.zeros(F * B, dtype=np.int32)), t(np.full(G, -1, dtype=np.int32)), t(np.zeros(F, dtype=np.int32))
    live_base = t(np.zeros(F * B, dtype=np.int32))
    part = t(np.zeros((n_parts, rmax, 2), dtype=np.float32))
    lat = _lib.Lattice(B, beam, F, ints["slen"].data_ptr(), ints["end_off"].data_ptr(), ints["nstart"].data_ptr(),
                       ints["nword"].data_ptr())
    st = _lib.BeamState(score.data_ptr(), lse.data_ptr(), None, bp.data_ptr(), node.data_ptr(), word.data_ptr(),
                        cnt.data_ptr(), live.data_ptr(), n_live.data_ptr(), edge.data_ptr(), live_base.data_ptr(), None, 0, 0)
    stream = _st() if cuda else 0
    for f in range(F):
        if f >= 1:
            if cuda:
                torch.cuda.synchronize()
            nl = int(n_live[f - 1].item())
            lv = live[(f - 1) * rmax:(f - 1) * rmax + nl].cpu().numpy().astype(np.int64)
            pn = np.zeros((n_parts, rmax, 2), dtype=np.float32)
            for q in range(n_parts):
                # slice maxima either equal or 200 below (exp underflows identically in f32 and f64): exact fold
                pn[q, :nl, 0] = 3.0 + (lv % 5) - 200.0 * ((lv + q) % 3 == 0)
                pn[q, :nl, 1] = 1.0 + ((lv + 3 * q) % 7) * 0.5
            pn[0, :nl, 0] = 3.0 + (lv % 5)
            part.copy_(t(pn))
            st.lse_part, st.ld_part, st.n_parts = part.data_ptr(), rmax, n_parts
        assert lib.jlm_beam_step(lat, st, f, 0, P["max_cands"], stream) == 0
    if cuda:
        torch.cuda.synchronize()
    out = dict(score=score, bp=bp, node=node, word=word, cnt=cnt, n_live=n_live, lse=lse)
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("B,beam,F,max_nodes,n_parts", [(7, 10, 9, 12, 24), (64, 3, 6, 40, 5), (3, 20, 22, 70, 96)])
def test_beam_step_fused_combine(L, B, beam, F, max_nodes, n_parts):
    P = _beam_problem(np.random.default_rng(B * 100 + beam + F), B, beam, F, max_nodes)
    want = _run_beam_fused(FK, P, False, n_parts)
    got = _run_beam_fused(L, P, True, n_parts)
    np.testing.assert_array_equal(got["cnt"], want["cnt"])
    rmax = P["rmax"]
    for f in range(F):
        for s in range(B):
            k = int(want["cnt"][f * B + s])
            sl = slice(f * rmax + s * beam, f * rmax + s * beam + k)
            for name in ("bp", "node", "word"):
                np.testing.assert_array_equal(got[name][sl], want[name][sl], err_msg="%s f=%d s=%d" % (name, f, s))
            np.testing.assert_allclose(got["score"][sl], want["score"][sl], rtol=0, atol=1e-9)
            if f < F - 1 and f < int(P["slen"][s]):
                np.testing.assert_allclose(got["lse"][sl], want["lse"][sl], rtol=0, atol=1e-9)


@pytest.mark.parametrize("R,C,sn", [(3, 2000, 0), (10, 50000, 0), (4, 301, 1)])
def test_softmax_rows(L, R, C, sn):
    rng = np.random.default_rng(R + C)
    ld = (C + 3) // 4 * 4
    y, yg = _pair(rng.standard_normal((R, ld)).astype(np.float32) * 3)
    p, pg = _pair(np.zeros((R, ld), dtype=np.float32))
    assert FK.jlm_softmax_rows(y.data_ptr(), p.data_ptr(), ld, R, C, sn, 0) == 0
    assert L.jlm_softmax_rows(yg.data_ptr(), pg.data_ptr(), ld, R, C, sn, _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(pg.cpu().numpy()[:, :C], p.numpy()[:, :C], rtol=3e-5, atol=1e-9)


def _pack_t(L, fmt):
    """the packer of the hypothesis rows for a row format: int8 planes (jlm_pack_t_mixed) or FP6 planes with block scales (ABI 11)"""
    return L.jlm_pack_t_mixed6 if fmt == "mx6" else L.jlm_pack_t_mixed


def _mixed_segments(L, rng, V, widths, bounds, b2_np, eT=10, fmt="int8"):
    """mixed rows of every segment (jlm_pack_mixed) + the scale arrays jlm_vocab_lse_mixed takes; fmt "mx6": s8 = 0 selects the
    FP6 cross-term planes (csrc/jlm_mx6_body.h)"""
    import ctypes
    n = len(widths)
    segs = (_lib.Segment * n)()
    ts, ds, s8 = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_float * n)()
    keep, Bs, off = [], [], 0
    b2 = torch.as_tensor(b2_np).cuda()
    for i, k in enumerate(widths):
        nv = bounds[i + 1] - bounds[i]
        B_np = (rng.standard_normal((nv, k)) * 0.08).astype(np.float32)
        Bs.append(B_np)
        Bg = torch.as_tensor(B_np).cuda()
        bmax = max(float(np.abs(B_np).max()), float(np.abs(b2_np[bounds[i]:bounds[i + 1]]).max()) * 1.4427 if k % 32 else 0.0)
        eB = int(np.floor(np.log2(2.0 ** 14 / bmax)))
        eT_i = eT
        if fmt == "mx6":        # as DeviceModel._build_mixed scales mx6 operands: eT + eB = 0 (descale = 1: the fixed-reference kernel forms run)
            eB, eT_i = 4, -4
        hmax = float(np.abs((B_np * np.float32(2.0 ** eB)).astype(np.float16).astype(np.float32)).max())
        s_b = 0.0 if fmt == "mx6" else 2.0 ** int(np.ceil(np.log2(hmax / 127.0)))
        nb = k // 32 if k % 32 == 0 else (k + 2 + 31) // 32          # a contraction that fills its last block: no bias columns
        dst = torch.zeros((nv, 32 * nb), dtype=torch.float32, device="cuda")
        assert L.jlm_pack_mixed(Bg.data_ptr(), nv, k, k, b2.data_ptr() + 4 * bounds[i], 2.0 ** eB, 2.0 ** eB * 1.4426950408889634,
                                s_b, dst.data_ptr(), 32 * nb, _st()) == 0
        keep += [Bg, dst]
        segs[i] = _lib.Segment(bounds[i], bounds[i + 1], k, off, dst.data_ptr(), 32 * nb)
        ts[i], ds[i], s8[i] = 2.0 ** eT_i, 2.0 ** -(eT_i + eB), s_b
        off += k
    return segs, ts, ds, s8, Bs, keep, off, b2


@pytest.mark.parametrize("widths,R", [([200, 100, 52], 75), ([256], 40), ([36, 128], 33)])
def test_mx6_packers_match_the_numpy_double(L, widths, R):
    """ABI 11: the device packers of the mx6 rows (csrc/jlm_mixed.hip pack_mx6_kernel / pack_t_mx6_kernel: f16 hi, FP6 planes of hi and of the
    residual with an E8M0 scale per 32 k-values, scale bytes in granule 7 of a row's first block) against tests/fake_hip.py's numpy
    restatement of the format -- byte for byte (round to nearest even on the e2m3 grid, the block exponent rule, the bit packing)."""
    import ctypes
    rng = np.random.default_rng(sum(widths) + R)
    V = 40 * len(widths) + 7
    bounds = [0] + [(i + 1) * (V // len(widths)) for i in range(len(widths) - 1)] + [V]
    b2_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
    b2_np[::5] = 0.0
    segs, ts, ds, s8, Bs, keep, ldt, b2 = _mixed_segments(L, rng, V, widths, bounds, b2_np, fmt="mx6")
    T_np = (np.tanh(rng.standard_normal((R, ldt))) * rng.uniform(1e-3, 1.0, size=(R, 1))).astype(np.float32)
    T_np[3] = 0.0                                        # an all-zero row: scale bytes 0, codes 0
    T_np[5, ::2] = 0.0
    T_np[7] *= np.float32(2.0 ** -30)
    T = torch.as_tensor(T_np).cuda()
    n = len(widths)
    ld_tm = L.jlm_mixed_t_stride(segs, n)
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), dtype=torch.float32, device="cuda")
    assert L.jlm_pack_t_mixed6(segs, ts, n, T.data_ptr(), ldt, None, R, None, Tm.data_ptr(), ld_tm, _st()) == 0
    torch.cuda.synchronize()
    # --- vocabulary rows: repack on the host from the same matrices
    off = 0
    for i, k in enumerate(widths):
        nv, nb = bounds[i + 1] - bounds[i], segs[i].ldb // 32
        dev_rows = keep[2 * i + 1].cpu().numpy().view(np.uint8).reshape(nv, nb * 128)
        host = np.zeros((nv, nb * 128), dtype=np.uint8)
        eB = 4                                           # (_mixed_segments' scale for mx6 rows)
        src = np.ascontiguousarray(Bs[i])
        bias = np.ascontiguousarray(b2_np[bounds[i]:bounds[i + 1]])
        assert FK.jlm_pack_mixed(src.ctypes.data, nv, k, k, bias.ctypes.data, 2.0 ** eB, 2.0 ** eB * 1.4426950408889634, 0.0,
                                 host.ctypes.data, 32 * nb, 0) == 0
        bad = np.argwhere(dev_rows != host)
        assert len(bad) == 0, ("vocabulary rows differ (row, byte):", i, bad[:8].tolist())
        off += nb * 128
    # --- hypothesis rows: the device image is granule-major in blocks of 32 rows (csrc/jlm_mixed_body.h); back to row-major
    host_tm = np.zeros((R, ld_tm * 4), dtype=np.uint8)
    tsc = [float(ts[i]) for i in range(n)]
    assert FK.jlm_pack_t_mixed6(segs, tsc, n, T_np.ctypes.data, ldt, None, R, None, host_tm.ctypes.data, ld_tm, 0) == 0
    raw = Tm.cpu().numpy().view(np.uint8).reshape(-1)
    dev_tm = np.zeros_like(host_tm)
    ngr = sum(segs[i].ldb // 32 for i in range(n)) * 8
    for r in range(R):
        blk = (r // 32) * 32 * ld_tm * 4
        for g in range(ngr):
            dev_tm[r, 16 * g:16 * g + 16] = raw[blk + g * 512 + (r % 32) * 16: blk + g * 512 + (r % 32) * 16 + 16]
    bad = np.argwhere(dev_tm[:, :16 * ngr] != host_tm[:, :16 * ngr])
    assert len(bad) == 0, ("hypothesis rows differ (row, byte):", bad[:8].tolist())


@pytest.mark.parametrize("V,widths,bounds,R,maxp", [(2000, [200, 100, 52], [0, 700, 1300, 2000], 48, 16), (2000, [200, 100, 52], [0, 700, 1300, 2000], 300, 96),
                                                    (3000, [256], [0, 3000], 200, 24), (1500, [512], [0, 1500], 130, 12)])
@pytest.mark.parametrize("fmt", ["int8", "mx6"])
def test_vocab_lse_mixed_identical_rows(L, V, widths, bounds, R, maxp, fmt):
    """Round 5: IDENTICAL hypothesis rows (frame 0 of every decode: all sentences leave the <eos> state) must get bit-identical slices
    whichever wave, row set or lane of the kernel they land in -- the order of a frame's live rows is not deterministic, so a
    kernel that rounds one row set differently from the other (the wide kernel did: one set's `s * scale + sum` was contracted
    into an fma, the other's was not) makes scores differ from run to run in their last bits.  Covers the eight-wave kernel, the
    wide kernel's two-row-set form (tied k = 256 by default; every shape under JLM_MX_WIDE=1) and its one-row-set form (k = 512)."""
    import ctypes
    if fmt == "mx6" and max(widths) > 256:
        pytest.skip("mx6 rows hold at most eight 32-k blocks")
    rng = np.random.default_rng(V + R)
    b2_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
    segs, ts, ds, s8, Bs, keep, ldt, b2 = _mixed_segments(L, rng, V, widths, bounds, b2_np, fmt=fmt)
    b2l = (b2 * 1.4426950408889634).contiguous()
    bias2 = b2l.data_ptr() if widths[0] % 32 == 0 else None
    T = torch.as_tensor(np.tile((np.tanh(rng.standard_normal((1, ldt))) * 0.7).astype(np.float32), (R, 1))).cuda()
    nd = torch.as_tensor(np.array([R], dtype=np.int32)).cuda()
    ld_tm = L.jlm_mixed_t_stride(segs, len(widths))
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), dtype=torch.float32, device="cuda")
    part = torch.zeros((96, R, 2), dtype=torch.float32, device="cuda")
    assert _pack_t(L, fmt)(segs, ts, len(widths), T.data_ptr(), ldt, None, R, nd.data_ptr(), Tm.data_ptr(), ld_tm, _st()) == 0
    lses = []
    for entry in (L.jlm_vocab_lse_mixed, L.jlm_vocab_lse_mixed_fr):       # (_fr: without a running maximum where the kernel has such a form)
        part.zero_()
        n = entry(segs, ds, s8, bias2, len(widths), Tm.data_ptr(), ld_tm, part.data_ptr(), R, maxp, R, nd.data_ptr(), _st())
        assert n >= len(widths), n
        torch.cuda.synchronize()
        p = part[:n].cpu().numpy()
        odd = np.argwhere((p != p[:, :1]).any(axis=2))
        assert len(odd) == 0, ("rows that differ from row 0 (slice, row):", odd[:8].tolist())
        q = p[:, 0].astype(np.float64)
        lses.append(np.log(np.sum(q[:, 1] * np.exp(q[:, 0] - q[:, 0].max()))) + q[:, 0].max())
    # the two forms differ in rounding only
    assert abs(lses[0] - lses[1]) <= 1e-6 * max(1.0, abs(lses[0])), lses


@pytest.mark.parametrize("V,widths,bounds,R", [(3000, [200, 100, 52], [0, 700, 1900, 3000], 300), (50000, [200, 100, 52], [0, 12000, 30000, 50000], 2560),
                                               (777, [60], [0, 777], 40), (5000, [252], [0, 5000], 513), (1000, [4, 36], [0, 300, 1000], 33),
                                               # contractions that fill their last block: biases from bias2 (tied k = 256; k = 128 / 64)
                                               (50000, [256], [0, 50000], 2560), (4001, [256], [0, 4001], 300), (3000, [128, 64], [0, 1700, 3000], 70),
                                               # k = 512 (an untied model's vocabulary matrix at H = 512): the wide kernel's one-row-set form, 128 rows per workgroup
                                               (4001, [512], [0, 4001], 300), (50000, [512], [0, 50000], 2560), (777, [512], [0, 777], 129)])
@pytest.mark.parametrize("fmt", ["int8", "mx6"])
def test_vocab_lse_mixed(L, V, widths, bounds, R, fmt):
    """jlm_vocab_lse_mixed (f16 hi.hi + int8 cross terms, csrc/jlm_mixed.hip): log-sum-exp of T.B^T + b2 over the vocabulary
    against the f64 evaluation of the f32 operands; the logits behind it are good to ~1e-5 of the row's logit scale (a one-word
    vocabulary range makes the kernel return a logit), the normaliser itself far better (the words' errors are independent)"""
    if fmt == "mx6" and max(widths) > 256:
        pytest.skip("mx6 rows hold at most eight 32-k blocks")
    rng = np.random.default_rng(V + R)
    b2_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
    segs, ts, ds, s8, Bs, keep, ldt, b2 = _mixed_segments(L, rng, V, widths, bounds, b2_np, fmt=fmt)
    b2l = (b2 * 1.4426950408889634).contiguous()
    bias2 = b2l.data_ptr() if widths[0] % 32 == 0 else None
    G = R + 9
    T_np = (np.tanh(rng.standard_normal((G, ldt))) * rng.uniform(0.05, 1.0, size=(G, 1))).astype(np.float32)
    T = torch.as_tensor(T_np).cuda()
    rows_np = rng.permutation(G)[:R].astype(np.int32)
    rows = torch.as_tensor(rows_np).cuda()
    nd = torch.as_tensor(np.array([R - 2], dtype=np.int32)).cuda()
    part = torch.zeros((96, R, 2), dtype=torch.float32, device="cuda")
    ld_tm = L.jlm_mixed_t_stride(segs, len(widths))
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), dtype=torch.float32, device="cuda")        # compact: packed row r = T row rows[r]
    assert _pack_t(L, fmt)(segs, ts, len(widths), T.data_ptr(), ldt, rows.data_ptr(), R, nd.data_ptr(), Tm.data_ptr(), ld_tm, _st()) == 0
    Tsel = T_np[rows_np[:R - 2]].astype(np.float64)
    y = np.concatenate([Tsel[:, segs[i].t_off:segs[i].t_off + widths[i]] @ Bs[i].astype(np.float64).T for i in range(len(widths))], axis=1) + b2_np
    ymax = y.max(axis=1)
    ref = ymax + np.log(np.exp(y - ymax[:, None]).sum(axis=1))
    # (the form without a running maximum too: on mx6 rows -- descale = 1 -- the D-softmax* and tied k = 256 shapes have kernel forms of their own)
    for entry in (L.jlm_vocab_lse_mixed, L.jlm_vocab_lse_mixed_fr):
        part.zero_()
        n = entry(segs, ds, s8, bias2, len(widths), Tm.data_ptr(), ld_tm, part.data_ptr(), R, 96, R, nd.data_ptr(), _st())
        assert n >= len(widths), n
        torch.cuda.synchronize()
        p = part[:n, :R - 2].cpu().numpy().astype(np.float64)
        with np.errstate(divide="ignore"):
            v = np.where(p[:, :, 1] > 0, p[:, :, 0] + np.log(p[:, :, 1]), -np.inf)
        mx = v.max(axis=0)
        lse = mx + np.log(np.exp(v - mx).sum(axis=0))
        assert np.abs(lse - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (entry.__name__ if hasattr(entry, "__name__") else entry, np.abs(lse - ref).max())
    # logits through one-word ranges (sampled words of every segment)
    worst = 0.0
    for i in range(len(widths)):
        for w in rng.choice(np.arange(bounds[i], bounds[i + 1]), size=min(6, bounds[i + 1] - bounds[i]), replace=False):
            w = int(w)
            nb = segs[i].ldb // 32
            one = (_lib.Segment * 1)(_lib.Segment(w, w + 1, widths[i], segs[i].t_off, segs[i].B + 128 * nb * (w - bounds[i]), segs[i].ldb))
            import ctypes
            ld1 = L.jlm_mixed_t_stride(one, 1)
            Tm1 = torch.zeros(((R + 31) // 32 * 32, ld1), dtype=torch.float32, device="cuda")
            assert _pack_t(L, fmt)(one, (ctypes.c_float * 1)(ts[i]), 1, T.data_ptr(), ldt, rows.data_ptr(), R, nd.data_ptr(),
                                   Tm1.data_ptr(), ld1, _st()) == 0
            n1 = L.jlm_vocab_lse_mixed(one, (ctypes.c_float * 1)(ds[i]), (ctypes.c_float * 1)(s8[i]), bias2, 1,
                                       Tm1.data_ptr(), ld1, part.data_ptr(), R, 96, R, nd.data_ptr(), _st())
            assert n1 == 1
            torch.cuda.synchronize()
            q = part[0, :R - 2].cpu().numpy().astype(np.float64)
            yw = q[:, 0] + np.log(q[:, 1])
            worst = max(worst, float((np.abs(yw - y[:, w]) / np.abs(y).max(axis=1)).max()))
    assert worst <= 3e-5, worst


@pytest.mark.parametrize("fmt", ["int8", "mx6"])
@pytest.mark.parametrize("tails", ["gauss", "outliers", "student-t3"])
def test_vocab_lse_mixed_spread(L, tails, fmt):
    """What the ONE int8 scale per segment costs on heavy-tailed vocabulary blocks: the error of the mixed-row normaliser grows with
    max|B| / rms B (the hi8 quantisation step is max|B| / 254 for every word) -- up to ~7e-7 x spread on the log-sum-exp; DeviceModel keeps
    blocks with a spread above JLM_MIXED_MAX_SPREAD (8) on split rows (jlm_amd/model.py, _build_mixed)."""
    rng = np.random.default_rng(7)
    V, k, R = 12000, 200, 256
    gen = {"gauss": lambda: rng.standard_normal((V, k)) * 0.08,
           "outliers": lambda: rng.standard_normal((V, k)) * 0.08 * np.where(rng.random((V, k)) < 1e-3, 30.0, 1.0),
           "student-t3": lambda: rng.standard_t(3, (V, k)) * 0.05}[tails]
    B_np = gen().astype(np.float32)
    spread = float(np.abs(B_np).max() / np.sqrt((B_np.astype(np.float64) ** 2).mean()))
    b2_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
    b2 = torch.as_tensor(b2_np).cuda()
    bmax = max(float(np.abs(B_np).max()), float(np.abs(b2_np).max()) * 1.4427)
    eB = int(np.floor(np.log2(2.0 ** 14 / bmax)))
    hmax = float(np.abs((B_np * np.float32(2.0 ** eB)).astype(np.float16).astype(np.float32)).max())
    s_b = 0.0 if fmt == "mx6" else 2.0 ** int(np.ceil(np.log2(hmax / 127.0)))
    nb = (k + 2 + 31) // 32
    Bg = torch.as_tensor(B_np).cuda()
    dst = torch.zeros((V, 32 * nb), dtype=torch.float32, device="cuda")
    assert L.jlm_pack_mixed(Bg.data_ptr(), V, k, k, b2.data_ptr(), 2.0 ** eB, 2.0 ** eB * 1.4426950408889634, s_b, dst.data_ptr(), 32 * nb, _st()) == 0
    import ctypes
    seg = (_lib.Segment * 1)(_lib.Segment(0, V, k, 0, dst.data_ptr(), 32 * nb))
    T_np = (np.tanh(rng.standard_normal((R, k))) * rng.uniform(0.05, 1.0, size=(R, 1))).astype(np.float32)
    T = torch.as_tensor(T_np).cuda()
    ld_tm = L.jlm_mixed_t_stride(seg, 1)
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), dtype=torch.float32, device="cuda")
    eT = 10
    assert _pack_t(L, fmt)(seg, (ctypes.c_float * 1)(2.0 ** eT), 1, T.data_ptr(), k, None, R, None, Tm.data_ptr(), ld_tm, _st()) == 0
    part = torch.zeros((96, R, 2), dtype=torch.float32, device="cuda")
    n = L.jlm_vocab_lse_mixed(seg, (ctypes.c_float * 1)(2.0 ** -(eT + eB)), (ctypes.c_float * 1)(s_b), None, 1, Tm.data_ptr(), ld_tm, part.data_ptr(),
                              R, 96, R, None, _st())
    assert n >= 1
    torch.cuda.synchronize()
    p = part[:n].cpu().numpy().astype(np.float64)
    v = p[:, :, 0] + np.log(p[:, :, 1])
    mx = v.max(axis=0)
    lse = mx + np.log(np.exp(v - mx).sum(axis=0))
    y = T_np.astype(np.float64) @ B_np.astype(np.float64).T + b2_np
    ymax = y.max(axis=1)
    ref = ymax + np.log(np.exp(y - ymax[:, None]).sum(axis=1))
    err = float(np.abs(lse - ref).max())
    if fmt == "mx6":
        # FP6 planes carry a scale per 32 k-values of every word: an outlier sets the scale of ITS block of 32 only, and the other 31 values
        # lose their low bits (outliers: 3e-5 at a spread of 80, student-t3: 1.1e-4 at a spread of several hundred -- inside what the int8
        # planes are allowed, 1e-6 x spread) -- the loader's calibration, not a spread gate, decides for this format
        print("mx6 spread", tails, spread, err)
        assert err <= (2e-6 if tails == "gauss" else 1e-6 * spread), (tails, spread, err)
        return
    assert err <= 1e-6 * spread, (tails, spread, err)
    if tails == "gauss":
        assert spread < 8 and err <= 2e-6, (spread, err)
    else:
        assert spread > 8, spread             # these are the blocks the loader keeps on split rows


@pytest.mark.parametrize("form", ["tail-split", "head-256", "first-split", "heads"])
@pytest.mark.parametrize("V,bounds,R", [(50000, [0, 12000, 30000, 50000], 2560), (3100, [0, 700, 1900, 3100], 300)])
def test_vocab_lse_hybrid(L, V, bounds, R, form):
    """jlm_vocab_lse_hybrid: the D-softmax* shapes in one launch -- k = 200 and k = 100 on mixed rows (int8 cross terms), k = 50 on
    split rows (three f16 passes) -- against the f64 evaluation of the f32 operands.  ABI 10: the head of a mixed segment on its split
    rows (`head-256`: the first 256 words of the first segment; `heads`: of the first two), a whole long segment on split rows beside
    mixed ones (`first-split`: what the loader falls back to for a model whose first segment carries the error)"""
    import ctypes
    mixed_set, heads = {"tail-split": ((0, 1), None), "head-256": ((0, 1, 2), [256, 0, 0]), "first-split": ((1, 2), None),
                        "heads": ((0, 1, 2), [128, 384, 0])}[form]
    widths = [200, 100, 52]
    rng = np.random.default_rng(V + R)
    b2_np = (rng.standard_normal(V) * 0.3).astype(np.float32)
    mx, ts, ds, s8, Bs, keep, ldt, b2 = _mixed_segments(L, rng, V, widths, bounds, b2_np)
    n = 3
    # the split view of every segment (the hybrid runs only the last one on it), from the same matrices
    plain = (_lib.Segment * n)()
    off = 0
    for i in range(n):
        Bg = keep[2 * i]
        plain[i] = _lib.Segment(bounds[i], bounds[i + 1], widths[i], off, Bg.data_ptr(), widths[i])
        off += widths[i]
    sp, sts, sds, bc, keep2 = _split_segments(L, plain, None, n, [6] * n, b2)
    for i in range(n):
        sts[i] = 2.0 ** 10
        sds[i] = 2.0 ** -(10 + 6)
    mixed = (_lib.Segment * n)()
    for i in mixed_set:
        mixed[i] = mx[i]
    nm = len(mixed_set)
    only = (_lib.Segment * nm)(*[mx[i] for i in mixed_set])
    G = R + 9
    T_np = (np.tanh(rng.standard_normal((G, ldt))) * rng.uniform(0.05, 1.0, size=(G, 1))).astype(np.float32)
    T = torch.as_tensor(T_np).cuda()
    rows_np = rng.permutation(G)[:R].astype(np.int32)
    rows = torch.as_tensor(rows_np).cuda()
    nd = torch.as_tensor(np.array([R - 2], dtype=np.int32)).cuda()
    ld_tm = L.jlm_mixed_t_stride(only, nm)
    Tm = torch.zeros(((R + 31) // 32 * 32, ld_tm), dtype=torch.float32, device="cuda")        # compact: packed row r = T row rows[r]
    assert L.jlm_pack_t_mixed(only, (ctypes.c_float * nm)(*[ts[i] for i in mixed_set]), nm, T.data_ptr(), ldt, rows.data_ptr(), R, nd.data_ptr(),
                              Tm.data_ptr(), ld_tm, _st()) == 0
    part = torch.zeros((100, R, 2), dtype=torch.float32, device="cuda")
    hs = (ctypes.c_int * n)(*heads) if heads else None
    nn = L.jlm_vocab_lse_hybrid(sp, sts, sds, bc, mixed, ds, s8, hs, n, b2.data_ptr(), T.data_ptr(), ldt, Tm.data_ptr(), ld_tm,
                                rows.data_ptr(), part.data_ptr(), R, 100, R, nd.data_ptr(), _st())
    assert nn >= n + (sum(1 for c in heads if c) if heads else 0), nn
    torch.cuda.synchronize()
    p = part[:nn, :R - 2].cpu().numpy().astype(np.float64)
    with np.errstate(divide="ignore"):
        v = np.where(p[:, :, 1] > 0, p[:, :, 0] + np.log(p[:, :, 1]), -np.inf)
    m = v.max(axis=0)
    lse = m + np.log(np.exp(v - m).sum(axis=0))
    Tsel = T_np[rows_np[:R - 2]].astype(np.float64)
    o = np.cumsum([0] + widths)
    y = np.concatenate([Tsel[:, o[i]:o[i + 1]] @ Bs[i].astype(np.float64).T for i in range(n)], axis=1) + b2_np
    ymax = y.max(axis=1)
    ref = ymax + np.log(np.exp(y - ymax[:, None]).sum(axis=1))
    assert np.abs(lse - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), np.abs(lse - ref).max()
