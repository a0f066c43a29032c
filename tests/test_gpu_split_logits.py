"""GPU: step logits of the SPLIT-f16 path -- the kernels the decode and the benchmark run (jlm_lstm_step_xg, jlm_gemm_nt_split,
jlm_vocab_lse_split) -- against the golden vectors captured from the reference (tests/golden/lm_steps.npz; reference
decoder/model.py:106-193), at the bar north_star sets: <= 1e-4 relative on the step logits.

The vocabulary kernel never forms logits.  Its debug view: over a ONE-word vocabulary range the log-sum-exp IS
that word's logit, so the production kernel -- same tile walk, same three f16 products, same fold -- is launched once per
sampled column with a one-word segment table.  The full-vocabulary launch is checked as well: the reference's logsumexp of a
row is y[col] - log(pred[col]) for any sampled column."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from jlm_amd import _lib, config as jconfig            # noqa: E402
from tests import golden_cases as gc                      # noqa: E402

pytestmark = pytest.mark.gpu
N_COLS = 48          # sampled columns per case (of the fixture's 256), spread over all segments


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _lse_of_parts(part, n_parts, rows):
    p = part[:n_parts, :rows].cpu().numpy().astype(np.float64)         # [parts, rows, 2] = (max, sum exp(y - max))
    with np.errstate(divide="ignore"):
        v = p[:, :, 0] + np.log(p[:, :, 1])
    v = np.where(p[:, :, 1] > 0, v, -np.inf)
    mx = v.max(axis=0)
    return mx + np.log(np.exp(v - mx).sum(axis=0))


@pytest.mark.parametrize("name", ["mid-vtable", "mid-tied", "small-vtable", "small-tied", "peaked20-vtable", "peaked20-tied", "heavy-vtable"])
def test_split_path_step_logits_within_1e4_of_reference(name, fx, golden_lm):
    f = fx(name)
    jconfig.set_root(f["root"])
    from jlm_amd.model import LSTM_Model
    lm = LSTM_Model(1)
    m = lm.dev
    assert m.split_lstm and m.split_array is not None, "the fixture must run the split-f16 decode path"
    L = _lib.lib()
    dev = m.device
    H, ldt = m.H, m.ldt
    worst = 0.0
    for rows in gc.LM_ROWS:
        idx, _subset, cols, h0, c0 = gc.lm_inputs(f["cfg"], rows)
        key = "%s/full/R%d" % (name, rows)
        steps = gc.LM_STEPS
        G = rows * (steps + 1)
        # state rows: block 0 = the golden initial state (h as split rows at the model's scale), block t + 1 = after step t
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
        # h', c' of the last step (split rows -> values)
        from tests.test_gpu_kernels import _unsplit
        h_last = (_unsplit(hs) / m.h_scale)[rows * steps:]
        c_last = cs[rows * steps:].cpu().numpy()
        np.testing.assert_allclose(h_last, golden_lm[key + "/h"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(c_last, golden_lm[key + "/c"], rtol=1e-4, atol=1e-5)
        # logits of sampled columns: the production vocabulary kernel over one-word ranges
        yref_all, pref_all = golden_lm[key + "/y"], golden_lm[key + "/pred"]
        pick = np.unique(np.linspace(0, len(cols) - 1, N_COLS).astype(int))
        part = torch.zeros((96, rows, 2), dtype=torch.float32, device=dev)
        y = np.zeros((rows, len(pick)))
        for j, ci in enumerate(pick):
            w = int(cols[ci])
            i = next(q for q, sg in enumerate(m.split_segments) if sg["v_start"] <= w < sg["v_end"])
            sg = m.split_segments[i]
            seg = (_lib.Segment * 1)(_lib.Segment(w, w + 1, sg["k"], sg["t_off"],
                                                  m.seg_split[i].data_ptr() + 4 * sg["ldb"] * (w - sg["v_start"]), sg["ldb"]))
            ts, ds = (ctypes.c_float * 1)(m.split_t_scale[i]), (ctypes.c_float * 1)(m.split_descale[i])
            bc = (ctypes.c_int * 1)(m.split_bias_col[i])
            n = L.jlm_vocab_lse_split(seg, ts, ds, bc, 1, m.b2.data_ptr(), T.data_ptr(), ldt, last.data_ptr(), part.data_ptr(),
                                      rows, 96, rows, None, _st())
            assert n >= 1
            torch.cuda.synchronize()
            y[:, j] = _lse_of_parts(part, n, rows)
        yref = yref_all[:, pick]
        scale = np.abs(yref_all).max(axis=1, keepdims=True)             # the row's logit scale, as the f32-pipe test uses
        rel = np.abs(y - yref) / scale
        worst = max(worst, float(rel.max()))
        assert rel.max() <= 1e-4, (key, rel.max())
        el = np.abs(y - yref) / np.maximum(np.abs(yref), 1.0)            # element-wise: |dy| <= 1e-4 max(|y|, 1)
        assert el.max() <= 1e-4, (key, el.max())
        # the full-vocabulary normaliser: logsumexp(y_row) = y[col] - log(pred[col]) in the reference
        n_seg = len(m.split_segments)
        segs = (_lib.Segment * n_seg)(*[_lib.Segment(sg["v_start"], sg["v_end"], sg["k"], sg["t_off"], m.seg_split[i].data_ptr(),
                                                     sg["ldb"]) for i, sg in enumerate(m.split_segments)])
        ts = (ctypes.c_float * n_seg)(*m.split_t_scale)
        ds = (ctypes.c_float * n_seg)(*m.split_descale)
        bc = (ctypes.c_int * n_seg)(*m.split_bias_col)
        n = L.jlm_vocab_lse_split(segs, ts, ds, bc, n_seg, m.b2.data_ptr(), T.data_ptr(), ldt, last.data_ptr(), part.data_ptr(),
                                  rows, 96, rows, None, _st())
        assert n >= 1
        torch.cuda.synchronize()
        lse = _lse_of_parts(part, n, rows)
        lse_ref = np.median(yref_all.astype(np.float64) - np.log(pref_all.astype(np.float64)), axis=1)
        assert np.abs(lse - lse_ref).max() <= 1e-4 * max(1.0, np.abs(lse_ref).max()), (key, np.abs(lse - lse_ref).max())
    print("%s: worst relative logit error of the split path %.2e" % (name, worst))
