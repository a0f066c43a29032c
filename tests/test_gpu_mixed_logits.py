"""GPU: the kernel that IS on the hot path -- jlm_vocab_lse_mixed (f16 hi.hi + int8 cross terms; csrc/jlm_mixed.hip), fed by
jlm_pack_t_mixed -- against the golden vectors captured from the reference (tests/golden/lm_steps.npz; reference
decoder/model.py:106-123, 141-193), on the Gaussian fixtures AND on the trained-model-like ones (peaked logits, heavy tails).

Same debug view as tests/test_gpu_split_logits.py: over a ONE-word vocabulary range the kernel's log-sum-exp is that word's logit,
so the production kernel is launched once per sampled column with a one-word segment table (its hypothesis rows packed for that
segment by the production packer); then the full-vocabulary launch against the reference's logsumexp(y) = y[col] - log(pred[col]).
The mixed rows are built here whatever the load-time gates would decide (JLM_MIXED_MAX_SPREAD / JLM_MIXED_MAX_LSE_RMS lifted): what
the gates decide is tested in test_mixed_row_gates_follow_the_models_logit_range below."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from jlm_amd import _lib, config as jconfig            # noqa: E402
from tests import golden_cases as gc                      # noqa: E402

pytestmark = pytest.mark.gpu
N_COLS = 48
FIXTURES = ["mid-vtable", "mid-tied", "peaked-vtable", "peaked20-vtable", "peaked-tied", "peaked20-tied", "heavy-vtable"]


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _lse_of_parts(part, n_parts, rows):
    p = part[:n_parts, :rows].cpu().numpy().astype(np.float64)
    with np.errstate(divide="ignore"):
        v = np.where(p[:, :, 1] > 0, p[:, :, 0] + np.log(p[:, :, 1]), -np.inf)
    mx = v.max(axis=0)
    return mx + np.log(np.exp(v - mx).sum(axis=0))


def _model(name, fx, monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    f = fx(name)
    jconfig.set_root(f["root"])
    from jlm_amd.model import LSTM_Model
    return f, LSTM_Model(1).dev


def _step_rows(m, L, cfg, rows):
    """the golden step sequence through the production step kernels -> (T [G, ldt], compact list of the last block's rows)"""
    idx, _subset, cols, h0, c0 = gc.lm_inputs(cfg, rows)
    steps, H, ldt, dev = gc.LM_STEPS, m.H, m.ldt, m.device
    G = rows * (steps + 1)
    hf = torch.zeros((G, H), dtype=torch.float32, device=dev)
    hf[:rows] = torch.as_tensor(h0, dtype=torch.float32)
    hs = torch.zeros_like(hf)
    assert L.jlm_pack_split_f16(hf.data_ptr(), rows, H, H, float(m.h_scale), hs.data_ptr(), H, _st()) == 0
    cs = torch.zeros((G, H), dtype=torch.float32, device=dev)
    cs[:rows] = torch.as_tensor(c0, dtype=torch.float32)
    T = torch.zeros((G, ldt), dtype=torch.float32, device=dev)
    prev = torch.arange(G, dtype=torch.int32, device=dev) - rows
    word = torch.zeros(G, dtype=torch.int32, device=dev)
    for t in range(steps):
        word[rows * (t + 1):rows * (t + 2)] = torch.as_tensor(idx[t], dtype=torch.int32)
    for t in range(steps):
        r = torch.arange(rows * (t + 1), rows * (t + 2), dtype=torch.int32, device=dev)
        assert L.jlm_lstm_step_xg(hs.data_ptr(), cs.data_ptr(), H, hs.data_ptr(), cs.data_ptr(), r.data_ptr(), prev.data_ptr(),
                                  word.data_ptr(), m.wt8.data_ptr(), m.xgate8.data_ptr(), H, float(m.gate_descale),
                                  float(m.h_scale), None, rows, None, _st()) == 0
        assert L.jlm_gemm_nt_split(hs.data_ptr(), H, r.data_ptr(), m.pmt_split.data_ptr(), H, None, T.data_ptr(), ldt,
                                   r.data_ptr(), None, float(m.t_descale), rows, m.pmt.shape[0], H, None, _st()) == 0
    torch.cuda.synchronize()
    last = torch.arange(rows * steps, rows * (steps + 1), dtype=torch.int32, device=dev)
    return T, last, cols


def _mixed_lse(m, L, segs, ts, ds, s8, bias2, T, last, rows, part):
    """pack the rows for exactly these segments, run the mixed normaliser on them -> log-sum-exp per row"""
    n = len(segs)
    arr = (_lib.Segment * n)(*segs)
    ld_tm = L.jlm_mixed_t_stride(arr, n)
    assert ld_tm > 0
    Tm = torch.zeros(((rows + 31) // 32 * 32, ld_tm), dtype=torch.float32, device=m.device)
    pack_t = L.jlm_pack_t_mixed6 if all(float(x) == 0.0 for x in s8) else L.jlm_pack_t_mixed       # (ABI 11: s8 = 0 -- FP6 planes)
    assert pack_t(arr, (ctypes.c_float * n)(*ts), n, T.data_ptr(), m.ldt, last.data_ptr(), rows, None, Tm.data_ptr(), ld_tm, _st()) == 0
    r = L.jlm_vocab_lse_mixed(arr, (ctypes.c_float * n)(*ds), (ctypes.c_float * n)(*s8), bias2, n, Tm.data_ptr(), ld_tm, part.data_ptr(),
                              rows, part.shape[0], rows, None, _st())
    assert r >= 1, r
    torch.cuda.synchronize()
    return _lse_of_parts(part, r, rows)


@pytest.mark.parametrize("fmt", ["mx6", "int8"])
@pytest.mark.parametrize("name", FIXTURES)
def test_mixed_kernel_step_logits_within_1e4_of_reference(name, fx, golden_lm, monkeypatch, fmt):
    """both formats of the cross-term planes: mx6 (FP6 with block scales on the block-scaled matrix instruction, round 6: the default) and
    int8 (JLM_LSE_MX6=0)"""
    f, m = _model(name, fx, monkeypatch, JLM_MIXED_MAX_SPREAD="1e30", JLM_MIXED_MAX_LSE_RMS="0", JLM_LSE_MIXED="1",
                  JLM_LSE_MX6="1" if fmt == "mx6" else "0")
    assert m.mixed_idx == list(range(m.n_segs)), "every segment of these fixtures has a mixed-row shape"
    assert m.mixed_fmt == fmt and all((float(x) == 0.0) == (fmt == "mx6") for x in m.mixed_s8)
    L = _lib.lib()
    bias2 = m.b2_log2.data_ptr() if m.b2_log2 is not None else None
    worst_rel, worst_abs, worst_lse = 0.0, 0.0, 0.0
    for rows in gc.LM_ROWS:
        T, last, cols = _step_rows(m, L, f["cfg"], rows)
        key = "%s/full/R%d" % (name, rows)
        yref_all, pref_all = golden_lm[key + "/y"], golden_lm[key + "/pred"]
        pick = np.unique(np.linspace(0, len(cols) - 1, N_COLS).astype(int))
        part = torch.zeros((128, rows, 2), dtype=torch.float32, device=m.device)
        y = np.zeros((rows, len(pick)))
        for j, ci in enumerate(pick):
            w = int(cols[ci])
            i = next(q for q, sg in enumerate(m.mixed_segments) if sg["v_start"] <= w < sg["v_end"])
            sg = m.mixed_segments[i]
            one = _lib.Segment(w, w + 1, sg["k"], sg["t_off"], m.seg_mixed[i].data_ptr() + 4 * sg["ldb"] * (w - sg["v_start"]), sg["ldb"])
            y[:, j] = _mixed_lse(m, L, [one], [m.mixed_t_scale[i]], [m.mixed_descale[i]], [m.mixed_s8[i]], bias2, T, last, rows, part)
        yref = yref_all[:, pick]
        scale = np.abs(yref_all).max(axis=1, keepdims=True)
        rel = np.abs(y - yref) / scale
        worst_rel = max(worst_rel, float(rel.max()))
        assert rel.max() <= 1e-4, (key, rel.max())
        # element-wise as well: |dy| <= 1e-4 max(|y|, 1)
        el = np.abs(y - yref) / np.maximum(np.abs(yref), 1.0)
        worst_abs = max(worst_abs, float(el.max()))
        assert el.max() <= 1e-4, (key, el.max())
        # the full-vocabulary launch: every segment, equal-cost columns, the fold across sub-ranges
        segs = [_lib.Segment(sg["v_start"], sg["v_end"], sg["k"], sg["t_off"], m.seg_mixed[i].data_ptr(), sg["ldb"])
                for i, sg in enumerate(m.mixed_segments)]
        lse = _mixed_lse(m, L, segs, m.mixed_t_scale, m.mixed_descale, m.mixed_s8, bias2, T, last, rows, part)
        lse_ref = np.median(yref_all.astype(np.float64) - np.log(pref_all.astype(np.float64)), axis=1)
        worst_lse = max(worst_lse, float(np.abs(lse - lse_ref).max()))
        assert np.abs(lse - lse_ref).max() <= 1e-4 * max(1.0, np.abs(lse_ref).max()), (key, np.abs(lse - lse_ref).max())
    print("%s [%s]: mixed kernel vs reference: logits %.2e of the row scale, %.2e element-wise; log-sum-exp %.2e absolute" % (
        name, fmt, worst_rel, worst_abs, worst_lse))


# fixture -> what the DEFAULT load keeps: (kept, format).  Round 6: the FP6 planes (mx6) are tried first; a model they do not pass on gets
# the int8 planes and their two-format launches as in round 5 (peaked20-vtable: second and third segment on int8 mixed rows, the first --
# which carries the mass and the error -- on split rows), then split rows.  Heavy-tailed blocks, which one int8 scale per segment kept on
# split rows, pass on mx6 (a scale per 32 k-values of every word).
GATES = [("mid-vtable", True, "mx6"), ("mid-tied", True, "mx6"), ("peaked-vtable", True, "mx6"), ("peaked-tied", True, "mx6"),
         ("peaked20-vtable", "first-split", "int8"), ("peaked20-tied", False, None), ("heavy-vtable", True, "mx6")]


@pytest.mark.parametrize("name,kept,fmt", GATES)
def test_mixed_row_gates_follow_the_models_logit_range(name, kept, fmt, fx, monkeypatch):
    """DeviceModel._build_mixed (format, spread of the blocks) and ._calibrate_mixed (the model's own log-normalisers in both forms on four
    probes -- uniform and 1 / rank word ids, two seeds -- at load): Gaussian, moderately peaked and heavy-tailed models keep a fast path,
    models whose logits reach +-20 fall back -- on the default settings, which is what every golden decode test runs with.  The decision
    must not hang on the probe: it is the same under three other seeds, and no fixture sits within 30 % of the limit."""
    for k in ("JLM_MIXED_MAX_SPREAD", "JLM_MIXED_MAX_LSE_RMS", "JLM_LSE_MIXED", "JLM_LSE_MX6"):
        monkeypatch.delenv(k, raising=False)
    from jlm_amd.model import DeviceModel
    seen = []
    for seed in (20240929, 7, 123456):
        monkeypatch.setattr(DeviceModel, "CALIB_SEED", seed)
        _f, m = _model(name, fx, monkeypatch)
        cal = m.mixed_calib
        print(name, "seed", seed, "fmt", m.mixed_fmt, "idx", m.mixed_idx, "spread", ["%.1f" % x for x in m.mixed_spread], "calibration", cal)
        seen.append((bool(m.mixed_idx), m.mixed_fmt, list(m.mixed_idx), None if cal is None else cal.get("margin")))
        assert bool(m.mixed_idx) == bool(kept), (name, m.mixed_spread, cal)
        assert m.mixed_fmt == fmt, (name, m.mixed_fmt, cal)
        assert cal is not None and cal["kept"] == bool(kept) and np.isfinite(cal["lse_rms_diff"]) and len(cal["probes"]) == len(DeviceModel.CALIB_PROBES)
        # no decision within 30 % of the limit (verdict round 5, item 3): the form kept is well inside, every form refused well outside
        assert cal["margin"] >= 1.3 or cal["margin"] <= 1.0 / 1.3, (name, cal)
        if "mx6" in cal:
            assert not cal["mx6"]["kept"] and cal["mx6"]["lse_rms_diff"] >= 1.3 * cal["limit"], cal
        if kept == "first-split":
            assert m.mixed_idx == [1, 2] and cal["split_segments"] == [0] and cal["lse_rms_diff"] < 2e-7 < 1e-6 < cal["lse_rms_diff_all_mixed"], cal
        if name == "mid-vtable":
            assert cal["lse_rms_diff"] < 5e-7, cal          # the headline model is far inside the limit
    assert len(set((a, b, tuple(c)) for a, b, c, _ in seen)) == 1, seen
