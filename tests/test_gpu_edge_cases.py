"""GPU: edge cases of the decoders against the oracle -- beam 1, beam larger than the
candidate count, one-kana and long sentences, raw-symbol (<unk>) fallbacks, empty
batch, a vocabulary whose size is not a multiple of any tile, LSTM_Model helpers."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from jlm_amd import config as jconfig, synth       # noqa: E402
from oracle import jlm_oracle as orc               # noqa: E402

pytestmark = pytest.mark.gpu


def _pair(f, kind):
    jconfig.set_root(f["root"])
    from jlm_amd.decoder import Decoder
    from jlm_amd.decoder_dynamic import DynamicDecoder
    d = (DynamicDecoder if kind == "dynamic" else Decoder)(1)
    o = (orc.OracleDynamicDecoder if kind == "dynamic" else orc.OracleDecoder)(f["root"], 1)
    return d, o


def _same(got, want, tag):
    assert len(got) == len(want), tag
    assert got[0][1] == want[0][1], (tag, got[0], want[0])
    np.testing.assert_allclose([s for s, _ in got], [s for s, _ in want], rtol=2e-6, atol=2e-5, err_msg=str(tag))


@pytest.mark.parametrize("kind", ["static", "static-vs", "dynamic"])
@pytest.mark.parametrize("beam,topN", [(1, 10), (2, 1), (64, 10), (33, 40)])
def test_beam_extremes(kind, beam, topN, fx):
    """(beams above 32 take two row blocks per group in the word-list kernels: static-vs and dynamic)"""
    f = fx("small-tied")
    d, o = _pair(f, "dynamic" if kind == "dynamic" else "static")
    kw = dict(vocab_select=True) if kind != "static" else {}
    sents = synth.make_ragged_sentences(7, 1, 14, seed=beam + 3, alphabet=f["alphabet"])
    got = d.decode_batch(sents, beam_width=beam, topN=topN, **kw)
    for s, g in zip(sents, got):
        _same(g, o.decode(s, beam_width=beam, topN=topN, **kw), (kind, beam, s))


@pytest.mark.parametrize("fixture,kind", [("small-vtable", "static"), ("small-tied", "static-vs"), ("small-tied", "dynamic"),
                                          ("small-untied", "static"), ("wide-vtable", "static"), ("wide128-tied", "static")])
@pytest.mark.parametrize("beam", [100, 257])
def test_beams_above_one_wave(fixture, kind, beam, fx):
    """beams above 64 (the reference has no limit, decoder.py:227-229): a lane of the sentence's wave owns several ranks in the
    beam step; the word-list kernels fall back to the forms without a beam limit"""
    f = fx(fixture)
    d, o = _pair(f, "dynamic" if kind == "dynamic" else "static")
    kw = dict(vocab_select=True) if kind != "static" else {}
    sents = synth.make_ragged_sentences(5, 2, 11, seed=beam, alphabet=f["alphabet"])
    got = d.decode_batch(sents, beam_width=beam, topN=beam, **kw)
    for s, g in zip(sents, got):
        w = o.decode(s, beam_width=beam, topN=beam, **kw)
        assert len(g) == len(w)
        _same(g, w, (kind, beam, s))


def test_dynamic_oversized_cells_take_the_host_path(fx, monkeypatch):
    """DynamicDecoder: sentences with a lattice cell above the device beam step's capacity go through the host-side search
    (DynamicDecoder._decode_host over the GPU predict kernels), the rest through the device -- both as the oracle"""
    f = fx("small-tied")
    d, o = _pair(f, "dynamic")
    sents = synth.make_ragged_sentences(9, 2, 10, seed=21, alphabet=f["alphabet"])
    from jlm_amd.decoder_dynamic import DynamicDecoder
    from jlm_amd.lattice import BatchLattice
    lat = BatchLattice(d._builder, sents, 6)
    per = np.diff(np.asarray(lat.end_off)).reshape(lat.n_frames, lat.n_sent).max(axis=0) * 6
    limit = int(np.sort(per)[len(per) // 2])
    monkeypatch.setattr(DynamicDecoder, "CAND_LIMIT", limit)
    assert 0 < int((per > limit).sum()) < len(sents)
    got = d.decode_batch(sents, beam_width=6, vocab_select=True)
    for s, g in zip(sents, got):
        _same(g, o.decode(s, beam_width=6, vocab_select=True), ("oversized-dynamic", s))


def test_beam_step_capacity_query_matches_the_launcher():
    """jlm_beam_step_max_cands is the launcher's own formula: the largest accepted max_cands passes, 256 more do not"""
    from jlm_amd import _lib
    L = _lib.lib()
    for beam, frames, mode in [(10, 24, 0), (64, 48, 2), (100, 24, 0), (1024, 16, 2)]:
        c = L.jlm_beam_step_max_cands(beam, frames, mode)
        assert c > 0 and c % 256 == 0
    assert L.jlm_beam_step_max_cands(2000, 8, 0) == 0


@pytest.mark.parametrize("kind", ["static", "dynamic"])
def test_one_kana_long_and_unknown_symbols(kind, fx):
    f = fx("small-vtable" if kind == "static" else "small-tied")
    d, o = _pair(f, kind)
    kw = dict(vocab_select=True) if kind == "dynamic" else {}
    a = f["alphabet"]
    long_s = synth.make_sentences(1, 90, seed=5, alphabet=a)[0]
    foreign = "".join(synth.KANA[c] for c in (40, 41, 3, 70, 2, 2, 55))        # symbols outside the 12-kana lexicon: <unk> nodes
    sents = [synth.KANA[0], long_s, foreign, synth.KANA[50], synth.KANA[1] * 30]
    got = d.decode_batch(sents, beam_width=8, **kw)
    for s, g in zip(sents, got):
        _same(g, o.decode(s, beam_width=8, **kw), (kind, s[:8]))
    assert any(len(w) == 1 and "/" not in w for _, ws in got[2] for w in ws)    # raw kana came through as words


def test_empty_batch_and_bad_input(fx):
    f = fx("small-tied")
    d, _ = _pair(f, "static")
    assert d.decode_batch([]) == []
    got = d.decode_batch(["アイ", ""])            # the reference's loop over an empty input leaves the <eos> path alone
    assert got[1] == [(0.0, [])] and got[0] == d.decode("アイ")
    assert d.decode("") == [(0.0, [])]
    with pytest.raises(ValueError):
        d.decode("アイ", beam_width=1025)      # JLM_MAX_BEAM (include/jlm_hip.h) is the widest beam of the device path
    with pytest.raises(ValueError):
        d.decode("アイ", beam_width=0)


def test_unpruned_search_and_stale_vocab_quirk(fx):
    """beam_width=None (decoder.py:227): every candidate survives, unsorted output -- the host-side path over the predict
    API against the oracle's restatement of the same loop; compat_quirks: the stale lattice_vocab of decoder.py:62,176"""
    f = fx("small-tied")
    d, o = _pair(f, "static")
    for s in synth.make_ragged_sentences(4, 2, 3, seed=21, alphabet=f["alphabet"]):
        want = orc.OracleDecoder(f["root"], 1).decode(s, beam_width=None)     # (a fresh oracle: it keeps the reference's stale list)
        d.lattice_vocab = None
        got = d.decode(s, beam_width=None)
        assert [w for _, w in got] == [w for _, w in want]             # generation order, not score order
        np.testing.assert_allclose([x for x, _ in got], [x for x, _ in want], rtol=2e-6, atol=2e-5)
        want = orc.OracleDecoder(f["root"], 1).decode(s, beam_width=None, vocab_select=True)
        got = d.decode(s, beam_width=None, vocab_select=True)
        assert [w for _, w in got] == [w for _, w in want]
        np.testing.assert_allclose([x for x, _ in got], [x for x, _ in want], rtol=2e-6, atol=2e-5)
    d.max_unpruned_paths = 50
    with pytest.raises(ValueError):
        d.decode("".join(synth.make_sentences(1, 12, seed=3, alphabet=f["alphabet"])), beam_width=None)
    # stale vocabulary: the oracle keeps the reference's behaviour (jlm_oracle.py: "stale lattice_vocab across calls")
    d2, o2 = _pair(f, "static")
    d2.compat_quirks = True
    s1, s2 = synth.make_ragged_sentences(2, 4, 6, seed=33, alphabet=f["alphabet"])
    assert [w for _, w in d2.decode(s1, beam_width=4, vocab_select=True)] == [w for _, w in o2.decode(s1, beam_width=4, vocab_select=True)]
    try:
        want = o2.decode(s2, beam_width=4)          # normalised over s1's list, or ValueError for a word outside it
    except ValueError:
        with pytest.raises(ValueError):
            d2.decode(s2, beam_width=4)
    else:
        got = d2.decode(s2, beam_width=4)
        assert [w for _, w in got] == [w for _, w in want]
        np.testing.assert_allclose([x for x, _ in got], [x for x, _ in want], rtol=2e-6, atol=2e-5)
    want = o2.decode(s1, beam_width=4)              # every word of s1 is in its own list: the quirk changes only the normaliser
    got = d2.decode(s1, beam_width=4)
    assert [w for _, w in got] == [w for _, w in want]
    np.testing.assert_allclose([x for x, _ in got], [x for x, _ in want], rtol=2e-6, atol=2e-5)


def test_odd_vocabulary_size(tmp_path):
    """V = 1237 (no tile divides it), H = 96, E = 20 (k padded to 4, single short k-step)."""
    root = str(tmp_path)
    cfg = synth.make_config(1237, 96, 20, "tied")
    lexicon, _rd = synth.write_lexicon(root, 1237, alphabet=10)
    synth.write_experiment(root, 1, cfg, scale=0.3)
    jconfig.set_root(root)
    from jlm_amd.decoder import Decoder
    d = Decoder(1)
    o = orc.OracleDecoder(root, 1)
    sents = synth.make_ragged_sentences(9, 2, 15, seed=8, alphabet=10)
    for s, g in zip(sents, d.decode_batch(sents, beam_width=7)):
        _same(g, o.decode(s, beam_width=7), s)
    # materialised logits of the same model (predict API) on an odd row count
    rng = np.random.RandomState(3)
    idx = [int(x) for x in rng.randint(0, 1237, size=13)]
    h0, c0 = rng.normal(0, 0.3, (13, 96)), rng.normal(0, 0.3, (13, 96))
    (pred, y, _a, _b), h, c = d.model.predict_with_context(idx, h0, c0)
    pr, yr, hr, cr, _, _ = o.model.predict(idx, h0, c0)
    assert np.abs(y - yr).max() <= 1e-4 * np.abs(yr).max()
    np.testing.assert_allclose(h, hr, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pred.sum(axis=1), 1.0, rtol=1e-4)


def test_model_helpers(fx):
    f = fx("small-tied")
    jconfig.set_root(f["root"])
    from jlm_amd.model import LSTM_Model
    lm = LSTM_Model(1)
    o = orc.OracleDecoder(f["root"], 1).model
    # project() on a 1-D hidden vector, float32 input
    hvec = np.random.RandomState(0).normal(0, 0.3, 64).astype(np.float32)
    y = lm.project(hvec, vocab=[5, 3, 9])
    yr = o.project(hvec[None].astype(np.float64), [5, 3, 9])
    assert np.abs(y - yr).max() <= 1e-4 * np.abs(yr).max()
    # reset=True keeps the previous row count (reference model.py:107-109)
    lm.predict([1, 2, 3])
    pred, _y, _t1, _t2 = lm.predict([4, 5, 6], reset=True)
    assert pred.shape[0] == 3
    # evaluate(): per-word -log p of a sequence (the reference's version is broken)
    nll = lm.evaluate(1, [7, 8, 9])
    h, c = o.zero_state(1)
    want, prev = [], 1
    for w in [7, 8, 9]:
        p, _yy, h, c, _, _ = o.predict([prev], h, c)
        want.append(-np.log(p[0, w]))
        prev = w
    np.testing.assert_allclose(nll, want, rtol=1e-4)


def _rescaled_fixture(tmp_path, name, edit):
    """a copy of fixture `name` whose weights went through edit(dict) -> same model family, other ranges"""
    import os
    import pickle
    root = str(tmp_path / name)
    cfg, _lx, _rd, alphabet = synth.build_fixture(root, name)
    wp = os.path.join(root, "train", "experiments", "1", "weights", "lstm_weights.pkl")
    with open(wp, "rb") as fh:
        w = pickle.load(fh)
    edit(w)
    with open(wp, "wb") as fh:
        pickle.dump(w, fh)
    return dict(root=root, cfg=cfg, alphabet=alphabet)


def _scale_out(k_out, k_proj):
    def edit(w):
        for key in list(w):
            if key.startswith("LM"):
                w[key] = [b * np.float32(k_out) for b in w[key]] if isinstance(w[key], list) else w[key] * np.float32(k_out)
            if key == "PM":
                w[key] = w[key] * np.float32(k_proj)
    return edit


def _big_bias(w):
    n = w["b2"].shape[0]
    w["b2"] = (w["b2"] + np.linspace(-40.0, 40.0, n).astype(np.float32)[np.random.RandomState(5).permutation(n)]).astype(np.float32)


def _zero_rows(w):
    for key in list(w):
        if key.startswith("LM") and not isinstance(w[key], list):
            w[key] = w[key].copy()
            w[key][3::7] = 0.0
    w["PM"] = w["PM"].copy()
    w["PM"][:, 1::5] = 0.0


@pytest.mark.parametrize("name,edit,tag", [
    ("small-vtable", _scale_out(16.0, 1.0 / 16.0), "large output embeddings, small projection"),
    ("small-tied", _scale_out(1.0 / 64.0, 64.0), "small embeddings, large projection"),
    ("small-vtable", _big_bias, "biases spread over +-40"),
    ("small-tied", _big_bias, "biases spread over +-40"),
    ("small-vtable", _scale_out(1.0, 1e-3), "projection ~1e-3: logits are the biases"),
    ("small-tied", _zero_rows, "zero embedding rows and projection columns"),
], ids=lambda v: v if isinstance(v, str) else getattr(v, "__name__", "edit"))
def test_operand_ranges_of_the_split_kernels(name, edit, tag, tmp_path):
    """The split-f16 kernels choose power-of-two scales from the weights' ranges (DeviceModel._build_split);
    the decode must stay on the oracle whatever those ranges are."""
    f = _rescaled_fixture(tmp_path, name, edit)
    d, o = _pair(f, "static")
    if os.environ.get("JLM_PRECISION", "f16x3") == "f16x3":      # (the suite also runs under JLM_PRECISION=f32)
        assert d.model.dev.split_array is not None
    sents = synth.make_ragged_sentences(6, 3, 15, seed=123, alphabet=f["alphabet"])
    got = d.decode_batch(sents, beam_width=6)
    for s, g in zip(sents, got):
        _same(g, o.decode(s, beam_width=6), (tag, s))


def _heavy_tails(w):
    rs = np.random.RandomState(9)
    for key in list(w):
        if key.startswith("LM") and not isinstance(w[key], list):
            w[key] = (w[key] * np.where(rs.rand(*w[key].shape) < 2e-3, 25.0, 1.0)).astype(np.float32)


@pytest.mark.parametrize("name,edit,tag,mixed", [
    ("wide-vtable", _scale_out(16.0, 1.0 / 16.0), "large output embeddings, small projection", True),
    ("wide-vtable", _big_bias, "biases spread over +-40", True),
    ("wide-vtable", _zero_rows, "zero embedding rows and projection columns", True),
    ("wide128-tied", _big_bias, "biases spread over +-40 (biases outside the rows)", True),
    ("wide128-tied", _scale_out(1.0 / 64.0, 64.0), "small embeddings, large projection", True),
    ("wide-vtable", _heavy_tails, "0.2 % of the embedding entries x 25: the spread guard keeps the model on split rows", False),
], ids=lambda v: v if isinstance(v, str) else getattr(v, "__name__", "edit") if callable(v) else str(v))
@pytest.mark.parametrize("fmt", ["mx6", "int8"])
def test_operand_ranges_of_the_mixed_kernels(name, edit, tag, mixed, tmp_path, monkeypatch, fmt):
    """The mixed-row normaliser takes its power-of-two scales (2^eB, the int8 scale s8, 2^eT) from the weights' ranges and only
    serves blocks without heavy tails (DeviceModel._build_mixed): the decode stays on the oracle whatever the ranges are.
    (The KERNELS' handling of the ranges is what is tested: the load-time calibration, which sends e.g. the +-40 biases to split
    rows on its own measurement, is switched off here -- tests/test_gpu_mixed_logits.py covers what it decides.)"""
    if os.environ.get("JLM_PRECISION", "f16x3") != "f16x3" or os.environ.get("JLM_LSE_MIXED", "1") == "0":
        pytest.skip("the suite is running without the mixed rows")
    monkeypatch.setenv("JLM_LSE_MX6", "1" if fmt == "mx6" else "0")
    heavy = edit is _heavy_tails
    if not (heavy and fmt == "mx6"):
        monkeypatch.setenv("JLM_MIXED_MAX_LSE_RMS", "0")
    # (round 6: the FP6 planes have no spread gate -- a scale per 32 k-values of every word -- so for the heavy-tailed model the
    #  load-time calibration stays ON in that format and decides; whatever it decides, the decode must stay on the oracle)
    f = _rescaled_fixture(tmp_path, name, edit)
    d, o = _pair(f, "static")
    if heavy and fmt == "mx6":
        print("heavy tails on mx6 rows:", d.model.dev.mixed_fmt, d.model.dev.mixed_idx, d.model.dev.mixed_calib)
    else:
        assert bool(d.model.dev.mixed_idx) == mixed, (d.model.dev.mixed_idx, d.model.dev.mixed_spread)
        assert d.model.dev.mixed_fmt == (fmt if mixed else None)
    sents = synth.make_ragged_sentences(6, 3, 15, seed=123, alphabet=f["alphabet"])
    got = d.decode_batch(sents, beam_width=6)
    for s, g in zip(sents, got):
        _same(g, o.decode(s, beam_width=6), (tag, s))


@pytest.mark.parametrize("mode", ["tied", "vtable", "untied"])
def test_kmeans_compressed_model_on_device(mode, tmp_path):
    """comp=8 (decoder/model.py:74-78, train/comp.py:52-80): the (code uint8, codebook) pairs go to the device, the vocabulary
    panels are expanded from them there (jlm_dequant_u8) and the decode equals the oracle run on the decoded weights"""
    from jlm_amd import weights as W
    root = str(tmp_path)
    cfg = synth.make_config(3000, 64, 32, mode, segs=[(32, 0, 700), (16, 700, 1800), (8, 1800, None)])
    synth.write_lexicon(root, 3000, alphabet=10)
    synth.write_experiment(root, 1, cfg, scale=0.3)
    decoded = synth.write_compressed(root, 1, bit=8, formats=("dump",))
    jconfig.set_root(root)
    from jlm_amd.decoder import Decoder
    d = Decoder(1, comp=8)
    m = d.model.dev
    assert len(m.seg_codes) == m.n_segs and all(c.dtype == torch.uint8 and c.is_cuda for c, _b in m.seg_codes.values())
    names = {"tied": ["LM"], "untied": ["UM"], "vtable": ["LM0", "LM1", "LM2"]}[mode]
    for i, nm in enumerate(names):          # bit-exact LUT expansion
        want = decoded[nm].T if mode == "untied" else decoded[nm]
        np.testing.assert_array_equal(m.seg_B[i].cpu().numpy()[:, :want.shape[1]], want.astype(np.float32))
    o = orc.OracleDecoder(root, 1)
    o.model = orc.OracleLM(o.config, W.load_weights(1, 8))          # what model.py:74-78 would have unpickled
    sents = synth.make_ragged_sentences(10, 2, 14, seed=4, alphabet=10)
    for s, g in zip(sents, d.decode_batch(sents, beam_width=6)):
        _same(g, o.decode(s, beam_width=6), (mode, s))


def test_oversized_lattice_cells_take_the_host_path(fx, monkeypatch):
    """sentences with a (frame, sentence) cell above the device beam step's LDS capacity: host-side beam search over the GPU
    predict kernels for those, the device frame loop for the rest -- both as the oracle"""
    f = fx("small-vtable")
    d, o = _pair(f, "static")
    sents = synth.make_ragged_sentences(9, 2, 10, seed=21, alphabet=f["alphabet"])
    from jlm_amd.decoder import Decoder
    from jlm_amd.lattice import BatchLattice
    lat = BatchLattice(d._builder, sents, 6)
    per = np.diff(np.asarray(lat.end_off)).reshape(lat.n_frames, lat.n_sent).max(axis=0) * 6
    limit = int(np.sort(per)[len(per) // 2])
    monkeypatch.setattr(Decoder, "CAND_LIMIT", limit)
    assert 0 < int((per > limit).sum()) < len(sents)
    got = d.decode_batch(sents, beam_width=6)
    for s, g in zip(sents, got):
        _same(g, o.decode(s, beam_width=6), ("oversized", s))


@pytest.mark.parametrize("name", ["wide-vtable"])          # (the D-softmax* shapes: a launch with a fixed-reference kernel form)
def test_fixed_reference_normaliser_overflow_is_loud(name, tmp_path, monkeypatch):
    """Round-5 advice / ABI 11: the normaliser without a running maximum (jlm_vocab_lse_mixed_fr: sum 2^y against the reference 0) on a
    model whose logits leave the f32 range.  The loader never enables the form for such a model (its probe bounds |log Z| at 40 bits);
    here it is FORCED on: an overflowed row's hypotheses score -inf and would be pruned silently -- the beam step raises the batch's
    flag word (jlm_beam_state.flags) and DecodeEngine.collect turns it into an error instead of a plausible n-best."""
    from jlm_amd import _lib
    if os.environ.get("JLM_PRECISION", "f16x3") != "f16x3" or os.environ.get("JLM_LSE_MIXED", "1") == "0":
        pytest.skip("the suite is running without the mixed rows")
    monkeypatch.setenv("JLM_MIXED_MAX_LSE_RMS", "0")          # no calibration: the rows stay mixed whatever the model
    f = _rescaled_fixture(tmp_path, name, _scale_out(600.0, 1.0))
    d, _o = _pair(f, "static")
    m = d.model.dev
    assert m.mixed_idx and not m.lse_fixed_ref and m.mixed_fmt == "mx6" and all(x == 1.0 for x in m.mixed_descale), (m.mixed_fmt, m.mixed_descale)
    sents = synth.make_ragged_sentences(4, 4, 9, seed=321, alphabet=f["alphabet"])
    ok = d.decode_batch(sents, beam_width=4)                   # the running-maximum form copes with any range
    assert all(np.isfinite(s) for r in ok for s, _ in r)
    m.lse_fixed_ref, m._decode_model = 1, None
    d._engine.m = m
    with pytest.raises(_lib.JlmHipError, match="not finite"):
        d.decode_batch(sents, beam_width=4)
