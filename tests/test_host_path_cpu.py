"""CPU-only: the real host code (weight packing, lattice CSR, engine launch
sequence, read-out, reference-compatible classes) driven end to end with the
numpy test double of the C ABI (tests/fake_hip.py), checked against the golden
vectors captured from the reference.  The kernels themselves are checked on the
GPU box (tests/test_gpu_*.py)."""
import os

import numpy as np
import pytest

from jlm_amd import config as jconfig, synth
from oracle import jlm_oracle as orc
from tests import fake_hip
from tests import golden_cases as gc

SMALL = [c for c in gc.DECODE_CASES if c[0].startswith("small-")]


@pytest.fixture()
def fake(monkeypatch):
    return fake_hip.install(monkeypatch)


def _decoder(f, kind):
    jconfig.set_root(f["root"])
    from jlm_amd.decoder import Decoder
    from jlm_amd.decoder_dynamic import DynamicDecoder
    d = (DynamicDecoder if kind == "dynamic" else Decoder)(1)
    d.perf_timing = False
    return d


@pytest.mark.parametrize("name", ["small-tied", "small-untied", "small-dsoftmax", "small-vtable", "small-tied-sn"])
def test_lstm_model_api_matches_reference(name, fx, fake, golden_lm):
    f = fx(name)
    jconfig.set_root(f["root"])
    from jlm_amd.model import LSTM_Model
    lm = LSTM_Model(1)
    for rows in gc.LM_ROWS:
        idx, subset, cols, h0, c0 = gc.lm_inputs(f["cfg"], rows)
        for kind in ("full", "subset"):
            if kind == "subset" and not f["cfg"]["share_embedding"]:
                continue
            vocab = subset if kind == "subset" else None
            h, c = h0.copy(), c0.copy()
            for step in range(gc.LM_STEPS):
                (pred, y, _t1, _t2), h, c = lm.predict_with_context(idx[step], h, c, vocab)
            key = "%s/%s/R%d" % (name, kind, rows)
            np.testing.assert_allclose(h, golden_lm[key + "/h"], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(c, golden_lm[key + "/c"], rtol=2e-5, atol=2e-6)
            ysel = y if kind == "subset" else y[:, cols]
            psel = pred if kind == "subset" else pred[:, cols]
            scale = np.abs(golden_lm[key + "/y"]).max()
            assert np.abs(ysel - golden_lm[key + "/y"]).max() <= 1e-4 * scale
            np.testing.assert_allclose(psel, golden_lm[key + "/pred"], rtol=2e-4)


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_decoders_match_reference_golden(case, fx, fake, golden_decode):
    name, fixture, kind, kwargs, spec = case
    f = fx(fixture)
    dec = _decoder(f, kind)
    dec.compat_quirks = gc.is_quirk_case(name)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_decode[name]
    if kwargs.get("random_sampling"):
        outs = []
        for si, s in enumerate(sents):
            np.random.seed(gc.RANDOM_SAMPLING_SEED + si)
            outs.append(dec.decode(s, **kwargs))
    else:
        outs = dec.decode_batch(sents, **kwargs)          # ragged batch in one go
    dyn_seg = kind == "dynamic" and fixture.split("-")[1] in ("dsoftmax", "vtable")
    assert dyn_seg == gc.is_quirk_case(name)          # the reference's result on segmented models needs the quirk mode
    for si, out in enumerate(outs):
        g = gold[si]["nbest"]
        assert len(out) == len(g), (name, si)
        assert [w for _, w in out] == [w for _, w in g], (name, si)
        np.testing.assert_allclose([sc for sc, _ in out], [sc for sc, _ in g], rtol=1e-5, atol=1e-4)


def test_single_sentence_api_and_attrs(fx, fake, golden_decode):
    f = fx("small-tied")
    dec = _decoder(f, "static")
    case = [c for c in gc.DECODE_CASES if c[0] == "small-tied/static-vs"][0]
    sents = gc.case_sentences(case[4], f["alphabet"])
    out = dec.decode(sents[2], **case[3])
    assert [w for _, w in out] == [w for _, w in golden_decode[case[0]][2]["nbest"]]
    assert dec.perf_sen == 1
    ends = orc.build_lattice(sents[2], dec.full_lexicon, dec.full_reading_dict, dec.w2i)
    # the reference's shape: dict frame -> [Node] (decoder.py:79-135)
    assert isinstance(dec.backward_lookup, dict) and sorted(dec.backward_lookup) == list(range(len(ends)))
    assert [[(n.start_idx, n.reading_length, n.word_idx, n.word) for n in dec.backward_lookup[i]] for i in range(len(ends))] == ends
    assert dec.lattice_vocab == orc.static_vocab(ends)
    bl = dec._build_lattice(sents[2])
    assert [[(n.start_idx, n.reading_length, n.word_idx, n.word) for n in bl[i]] for i in range(len(ends))] == ends
    assert dec._check_oov("no-such-word") and not dec._check_oov("<eos>")


def test_dynamic_requires_vocab_select(fx, fake):
    f = fx("small-tied")
    dec = _decoder(f, "dynamic")
    with pytest.raises(TypeError):
        dec.decode("アイウ", vocab_select=False)
    # round 6: beam_width=None is the reference's unpruned search (decoder_dynamic.py:89-91), on the host path -- it still needs the
    # vocabulary lists; the golden case small-tied/dynamic-unpruned checks what it returns
    with pytest.raises(TypeError):
        dec.decode("アイウ", beam_width=None, vocab_select=False)
    assert len(dec.decode("アイ", beam_width=None, vocab_select=True)) >= 1
    with pytest.raises(ValueError):
        dec.decode("アイウ", beam_width=1025, vocab_select=True)
    with pytest.raises(ValueError):
        dec.decode("アイウ", beam_width=0, vocab_select=True)


def test_dynamic_lattice_vocab_matches_reference_lists(fx, fake):
    f = fx("small-tied")
    dec = _decoder(f, "dynamic")
    o = orc.OracleDynamicDecoder(f["root"], 1)
    for s in gc.case_sentences(("ragged", 5, 2, 12, 21), f["alphabet"]):
        dec.decode(s, vocab_select=True, samples=7, top_sampling=True)
        o.decode(s, vocab_select=True, samples=7, top_sampling=True)
        assert dec.lattice_vocab == o.lattice_vocab


@pytest.mark.parametrize("kind,kw", [("static", {}), ("static", {"vocab_select": True}), ("dynamic", {"vocab_select": True})])
def test_chunked_pipeline_equals_one_batch(kind, kw, fx, fake):
    """decode_batch over many chunks (lattices prefetched by several worker threads, chunks in flight on
    alternating plans, results in input order) returns what one big batch returns."""
    f = fx("small-tied")
    dec = _decoder(f, kind)
    sents = gc.case_sentences(("ragged", 23, 1, 14, 77), f["alphabet"])
    whole = dec.decode_batch(sents, beam_width=5, **kw)
    lv_whole = dec.lattice_vocab
    dec.max_batch, dec.prefetch_workers = 4, 3
    def same(a, b):            # the numpy double's BLAS sums depend on the batch shape in the last bits
        assert [[w for _, w in x] for x in a] == [[w for _, w in x] for x in b]
        for x, y in zip(a, b):
            np.testing.assert_allclose([v for v, _ in x], [v for v, _ in y], rtol=1e-7)
    chunked = dec.decode_batch(sents, beam_width=5, **kw)
    same(chunked, whole)
    if kw:
        assert dec.lattice_vocab == lv_whole           # the last sentence's lists, as the reference leaves them
    dec.prefetch_workers = 1
    same(dec.decode_batch(sents, beam_width=5, **kw), whole)


def test_module_helpers_match_the_oracle_and_smoke_main_runs(fx, fake, capsys):
    """decoder/model.py:12-33 helpers (host side, for predict()'s numpy outputs) and its __main__ smoke (model.py:213-245)."""
    from jlm_amd import model as jm
    rng = np.random.RandomState(3)
    x = rng.randn(4, 37) * 5
    np.testing.assert_allclose(jm.softmax(x), orc.softmax(x), rtol=1e-12)
    assert jm.softmax(x[0]).shape == (1, 37)
    np.testing.assert_allclose(jm.sigmoid(x), orc.sigmoid(x), rtol=1e-12)
    np.testing.assert_allclose(jm.tanh(x), np.tanh(x))
    assert jm.find_top_N(np.array([0.1, 0.5, 0.2, 0.9]), 2).tolist() == [3, 1]
    np.random.seed(5)
    draws = [jm.sample([0.05, 0.9, 0.05]) for _ in range(50)]
    assert max(set(draws), key=draws.count) == 1 and set(draws) <= {0, 1, 2}
    np.random.seed(7)
    assert jm.sample([0.2, 0.3, 0.5], temperature=1e-3) == 2         # a cold temperature is an arg-max
    f = fx("small-tied")
    np.random.seed(11)
    a, b = jm.main(["-e", "1", "--root", f["root"], "--steps", "12"])
    out = capsys.readouterr().out
    assert "--- generated sentence" in out and "--- random sentence" in out
    assert np.isfinite(a) and np.isfinite(b) and a > 0 and b > 0


def test_usable_cpus_follows_quota_and_local_world_size(monkeypatch):
    import jlm_amd
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    base = jlm_amd.usable_cpus()
    assert 1 <= base <= (os.cpu_count() or 1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert jlm_amd.usable_cpus() == max(1, base // 4)           # the ranks of a node share the quota
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "not a number")
    assert jlm_amd.usable_cpus() == base


def test_unpruned_search_and_compat_quirks_cpu(fx, fake):
    """beam_width=None (decoder.py:227) and the stale lattice_vocab (decoder.py:62,176, compat_quirks) through the host-side
    path over the predict API, against the oracle's restatement of the reference loop"""
    from jlm_amd import synth
    f = fx("small-tied")
    d = _decoder(f, "static")
    o = orc.OracleDecoder(f["root"], 1)
    for s in synth.make_ragged_sentences(3, 2, 3, seed=21, alphabet=f["alphabet"]):
        for kw in ({}, dict(vocab_select=True)):
            want = orc.OracleDecoder(f["root"], 1).decode(s, beam_width=None, **kw)
            d.lattice_vocab = None
            got = d.decode(s, beam_width=None, **kw)
            assert [w for _, w in got] == [w for _, w in want]         # generation order, not score order
            np.testing.assert_allclose([x for x, _ in got], [x for x, _ in want], rtol=2e-6, atol=2e-5)
    assert d.decode("", beam_width=None) == [(0.0, [])]
    d2 = _decoder(f, "static")
    d2.compat_quirks = True
    s1 = synth.make_ragged_sentences(1, 4, 6, seed=33, alphabet=f["alphabet"])[0]
    assert [w for _, w in d2.decode(s1, beam_width=4, vocab_select=True)] == [w for _, w in o.decode(s1, beam_width=4, vocab_select=True)]
    want = o.decode(s1, beam_width=4)           # the oracle keeps the stale list, as the reference does
    got = d2.decode(s1, beam_width=4)
    assert [w for _, w in got] == [w for _, w in want]
    np.testing.assert_allclose([x for x, _ in got], [x for x, _ in want], rtol=2e-6, atol=2e-5)
    d2.compat_quirks = False                    # default: a full-vocabulary call normalises over the full vocabulary
    fresh = orc.OracleDecoder(f["root"], 1).decode(s1, beam_width=4)
    np.testing.assert_allclose([x for x, _ in d2.decode(s1, beam_width=4)], [x for x, _ in fresh], rtol=2e-6, atol=2e-5)


def test_decode_batch_chunks_by_length_and_memory_budget(fx, fake):
    """decode_batch deals sentences into device batches by decreasing length, closes a batch at max_batch sentences or at
    plan_budget_bytes of state rows, and hands the results back in the caller's order"""
    from jlm_amd import synth
    f = fx("small-tied")
    d = _decoder(f, "static")
    sents = synth.make_ragged_sentences(30, 1, 12, seed=5, alphabet=f["alphabet"]) + [synth.make_sentences(1, 60, seed=6, alphabet=f["alphabet"])[0]]
    d.max_batch = 8
    chunks = d._chunks(sents, 5)
    assert sorted(i for c in chunks for i in c) == list(range(len(sents))) and all(len(c) <= 8 for c in chunks)
    assert chunks[0][0] == len(sents) - 1                                  # the 60-kana outlier leads the first batch
    lens = [[len(sents[i]) for i in c] for c in chunks]
    assert all(l == sorted(l, reverse=True) for l in lens) and lens[1][0] >= lens[-1][0]
    m = d.model.dev
    row = (2 * m.H + m.ldt) * 4 + 64
    d.plan_budget_bytes = 64 * 2 * 5 * row                                 # room for two sentences of 64 frames at beam 5
    tight = d._chunks(sents, 5)
    assert len(tight[0]) <= 2 and sorted(i for c in tight for i in c) == list(range(len(sents)))
    assert d._chunks(sents, 5, reorder=False)[0][0] == 0                   # random_sampling: the caller's order is kept
    want = [d.decode(s, beam_width=5) for s in sents]
    got = d.decode_batch(sents, beam_width=5)
    for a, b in zip(got, want):
        assert [w for _, w in a] == [w for _, w in b]
        np.testing.assert_allclose([x for x, _ in a], [x for x, _ in b], rtol=1e-9, atol=1e-6)


def test_oversized_lattice_cells_take_the_host_path(fx, fake, monkeypatch):
    """a cell with more candidates than the device beam step keeps in LDS: those sentences go through the host-side beam search
    (Decoder._decode_unpruned with the beam), the rest of the batch through the device, results as the oracle's"""
    f = fx("small-tied")
    dec = _decoder(f, "static")
    sents = synth.make_ragged_sentences(9, 2, 10, seed=21, alphabet=f["alphabet"])
    want = dec.decode_batch(sents, beam_width=6)
    from jlm_amd.decoder import Decoder
    from jlm_amd.lattice import BatchLattice
    import numpy as np
    lat = BatchLattice(dec._builder, sents, 6)
    per = np.diff(np.asarray(lat.end_off)).reshape(lat.n_frames, lat.n_sent).max(axis=0) * 6
    limit = int(np.sort(per)[len(per) // 2])          # about half of the sentences are "too large"
    monkeypatch.setattr(Decoder, "CAND_LIMIT", limit)
    assert 0 < int((per > limit).sum()) < len(sents)
    got = dec.decode_batch(sents, beam_width=6)
    for g, w in zip(got, want):
        assert [x for _, x in g] == [x for _, x in w]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in w], rtol=1e-5, atol=1e-4)
    got_vs = dec.decode_batch(sents, beam_width=6, vocab_select=True)
    monkeypatch.setattr(Decoder, "CAND_LIMIT", 13000)
    want_vs = dec.decode_batch(sents, beam_width=6, vocab_select=True)
    for g, w in zip(got_vs, want_vs):
        assert [x for _, x in g] == [x for _, x in w]
    # the incremental decoder: the same routing, its own host-side search (DynamicDecoder._decode_host)
    from jlm_amd.decoder_dynamic import DynamicDecoder
    dyn = _decoder(f, "dynamic")
    want_dyn = dyn.decode_batch(sents, beam_width=6, vocab_select=True)
    monkeypatch.setattr(DynamicDecoder, "CAND_LIMIT", limit)
    got_dyn = dyn.decode_batch(sents, beam_width=6, vocab_select=True)
    for g, w in zip(got_dyn, want_dyn):
        assert [x for _, x in g] == [x for _, x in w]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in w], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name,kind,kw", [("small-tied", "static", {}), ("small-vtable", "static", {}),
                                          ("small-tied", "static", dict(vocab_select=True)),
                                          ("small-tied", "dynamic", dict(vocab_select=True))])
def test_beams_above_one_wave(fx, fake, name, kind, kw):
    """beam 100 (the reference has no limit, decoder.py:227-229): the beam step gives a lane several ranks; results as the
    oracle's"""
    from oracle import jlm_oracle as orc
    f = fx(name)
    dec = _decoder(f, kind)
    o = (orc.OracleDynamicDecoder if kind == "dynamic" else orc.OracleDecoder)(f["root"], 1)
    sents = synth.make_ragged_sentences(4, 3, 9, seed=77, alphabet=f["alphabet"])
    got = dec.decode_batch(sents, beam_width=100, topN=100, **kw)
    for s, g in zip(sents, got):
        w = o.decode(s, beam_width=100, topN=100, **kw)
        assert len(g) == len(w) and len(g) > 10
        assert [x for _, x in g][0] == [x for _, x in w][0]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in w], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("fmt", ["mx6", "int8"])
@pytest.mark.parametrize("name", ["wide-vtable", "wide-dsoftmax", "wideh-vtable", "wide128-tied"])
def test_mixed_rows_normaliser_route(fx, fake, monkeypatch, name, fmt):
    """Segments of width 200 / 100 / 50 get mixed rows at load (DeviceModel._build_mixed), the plan its packed-row buffer, and the
    frame loop packs the live rows and calls the hybrid normaliser (include/jlm_hip.h ABI 7); results as the oracle's, and as
    the split rows' (JLM_LSE_MIXED=0).  Round 6: in both formats of the cross-term planes -- mx6 (FP6 with block scales, ABI 11: the
    default where every segment has a hosted shape) and int8 (JLM_LSE_MX6=0; also what a model with a split-row segment gets)."""
    f = fx(name)
    calls = {"hybrid": 0, "pack_t": 0}
    lib = fake
    monkeypatch.setenv("JLM_CALIB_PATHS", "0")        # (the launch counts below are those of the model's own probes; the decoder's extra
    #                                                    probe from decoded paths has a test of its own)
    if fmt == "int8":
        monkeypatch.setenv("JLM_LSE_MX6", "0")
    want_fmt = "int8" if name.startswith("wideh") else fmt          # (the two-format launch hosts int8 planes only)
    # wide-*: all three widths are hosted mixed shapes, the all-mixed launch; wideh-*: the last one stays on split rows, the hybrid
    launch = "jlm_vocab_lse_hybrid" if name.startswith("wideh") else "jlm_vocab_lse_mixed"
    packer = "jlm_pack_t_mixed6" if want_fmt == "mx6" else "jlm_pack_t_mixed"
    hy, pk = getattr(lib, launch), getattr(lib, packer)

    def hybrid(*a):
        calls["hybrid"] += 1
        return hy(*a)

    def pack_t(*a):
        calls["pack_t"] += 1
        return pk(*a)
    monkeypatch.setattr(lib, launch, hybrid, raising=False)
    monkeypatch.setattr(lib, packer, pack_t, raising=False)
    dec = _decoder(f, "static")
    m = dec.model.dev
    # the load-time calibration (DeviceModel._calibrate_mixed) has run the normaliser once in each form on its probe rows -- and once
    # more in the fixed-reference form where the model's range allows it (round 6: the form itself is probed before it is enabled)
    n_probe = len(m.CALIB_PROBES) * (1 + int("fixed_ref_lse_rms_diff" in m.mixed_calib))
    assert len(m.mixed_calib["probes"]) == len(m.CALIB_PROBES) and m.mixed_calib["lse_rms_diff"] == max(x["rms"] for x in m.mixed_calib["probes"])
    assert calls["hybrid"] == calls["pack_t"] == n_probe and m.mixed_calib["kept"] and m.mixed_calib["lse_rms_diff"] < 1e-6, m.mixed_calib
    assert m.mixed_fmt == want_fmt == m.mixed_calib["fmt"] and all((x == 0.0) == (want_fmt == "mx6") for x in m.mixed_s8), (m.mixed_fmt, m.mixed_s8)
    if m.lse_fixed_ref:
        assert m.mixed_calib["fixed_ref_lse_rms_diff"] < 1e-6
    calls["hybrid"] = calls["pack_t"] = 0
    if name.startswith("wideh"):
        assert m.mixed_idx == [0, 1] and m.ld_tm == (7 * 128 + 4 * 128 + 32) // 4
        assert [sg["ldb"] for sg in m.mixed_segments] == [224, 128]
    elif name == "wide128-tied":      # k = 128 fills its four blocks: no bias columns, the biases travel as b2_log2
        assert m.mixed_idx == [0] and m.ld_tm == (4 * 128 + 32) // 4 and m.mixed_segments[0]["ldb"] == 128
        assert m.b2_log2 is not None
    else:
        assert m.mixed_idx == [0, 1, 2] and m.ld_tm == (7 * 128 + 4 * 128 + 2 * 128 + 32) // 4
        assert [sg["ldb"] for sg in m.mixed_segments] == [224, 128, 64]
    sents = synth.make_ragged_sentences(6, 2, 9, seed=5, alphabet=f["alphabet"])
    got = dec.decode_batch(sents, beam_width=8)
    n_steps = max(len(s) for s in sents)             # frames 0 .. L - 1 are stepped
    assert calls["hybrid"] == calls["pack_t"] == n_steps
    o = orc.OracleDecoder(f["root"], 1)
    for s, g in zip(sents, got):
        w = o.decode(s, beam_width=8)
        assert [x for _, x in g] == [x for _, x in w]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in w], rtol=1e-6, atol=2e-5)
    # the split rows alone: the same n-best
    monkeypatch.setenv("JLM_LSE_MIXED", "0")
    dec0 = _decoder(f, "static")
    assert dec0.model.dev.mixed_idx == [] and dec0.model.dev.ld_tm == 0
    before = calls["hybrid"]
    got0 = dec0.decode_batch(sents, beam_width=8)
    assert calls["hybrid"] == before
    for g, g0 in zip(got, got0):
        assert [x for _, x in g] == [x for _, x in g0]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in g0], rtol=1e-6, atol=2e-5)
    # heavy-tailed blocks stay on split rows: the guard compares max|B| / rms B with JLM_MIXED_MAX_SPREAD (Gaussian-like fixtures: ~4-5)
    assert m.mixed_spread and all(3.0 < x < 8.0 for x in m.mixed_spread), m.mixed_spread
    monkeypatch.delenv("JLM_LSE_MIXED")
    monkeypatch.setenv("JLM_MIXED_MAX_SPREAD", "2")
    dec2 = _decoder(f, "static")
    if want_fmt == "mx6":        # a scale per 32 k-values of every word: the spread gate does not apply to this format
        assert dec2.model.dev.mixed_fmt == "mx6" and dec2.model.dev.mixed_idx == m.mixed_idx
    else:
        assert dec2.model.dev.mixed_idx == [] and dec2.model.dev.ld_tm == 0 and dec2.model.dev.mixed_spread


def test_lattice_blocks_return_to_the_pool(fx, fake):
    """Lattices of a pipelined decode are built into pooled blocks (lattice.StagingPool; page-locked on a GPU box) and copied to the
    device from there; a block goes back when its batch has been read out, the last chunk's lattice (Decoder.last_lattice, what
    decode() builds backward_lookup from) keeps its arrays; JLM_PINNED_LATTICE=0 gives the same results through the staging copy."""
    f = fx("small-tied")
    dec = _decoder(f, "static")
    pool = dec._engine.staging_pool
    assert pool is not None
    sents = synth.make_ragged_sentences(40, 2, 9, seed=11, alphabet=f["alphabet"])
    dec.max_batch = 8                                   # five chunks
    a = dec.decode_batch(sents, beam_width=5)
    n_free = sum(len(v) for v in pool._free.values())
    assert pool.allocated >= 2 and n_free == pool.allocated - 1          # every block but the last lattice's is back
    assert dec.last_lattice.node_word is not None and dec.last_lattice._block is not None
    dec.last_lattice.backward_lookup(0)                 # its arrays are alive
    b = dec.decode_batch(sents, beam_width=5)           # blocks are reused
    assert pool.allocated <= n_free + 2
    dec._engine.staging_pool = None                     # the staging-copy path
    c = dec.decode_batch(sents, beam_width=5)
    for x, y, z in zip(a, b, c):
        assert x == y == z


def test_chunks_in_flight_rule(fx, fake):
    """Decoder.depth_for: one chunk per launch stream in flight, three at most for chunks above 8 192 hypothesis rows."""
    dec = _decoder(fx("small-tied"), "static")
    n = dec._engine.n_streams
    assert dec.pipeline_depth == n
    assert dec.depth_for(256, 10) == n and dec.depth_for(1024, 8) == n
    assert dec.depth_for(1024, 20) == min(n, 3) and dec.depth_for(8192, 10) == min(n, 3)


def test_packed_rows_take_32_blocks_at_most(tmp_path, fake):
    """jlm_pack_t_mixed holds a hypothesis row's 16-value groups in the 64 lanes of one wave: segment tables above 32 blocks per row are
    rejected by the real library (-2: jlm_mixed_t_stride, jlm_pack_t_mixed -- host-side checks, no GPU needed) and by its numpy
    double, and DeviceModel keeps such a model on split rows (five segments of k = 200: 35 blocks)."""
    import ctypes
    from jlm_amd import _lib
    real = ctypes.CDLL(_lib.LIB_PATH)
    mk = lambda n: (_lib.Segment * n)(*[_lib.Segment(100 * i, 100 * (i + 1), 200, 200 * i, None, 224) for i in range(n)])
    for lib_ in (real, fake):
        assert lib_.jlm_mixed_t_stride(mk(4), 4) == (4 * 7 * 128 + 32) // 4
        assert lib_.jlm_mixed_t_stride(mk(5), 5) == -2
    ts = (ctypes.c_float * 5)(*[1.0] * 5)
    real.jlm_pack_t_mixed.restype = ctypes.c_int
    assert real.jlm_pack_t_mixed(mk(5), ts, 5, None, 1000, None, 4, None, None, (5 * 7 * 128 + 32) // 4, None) == -2
    V = 1000
    segs = [(200, 200 * i, 200 * (i + 1) if i < 4 else None) for i in range(5)]
    cfg = synth.make_config(V, 64, 200, "vtable", segs)
    synth.write_lexicon(str(tmp_path), V, alphabet=12)
    synth.write_experiment(str(tmp_path), 1, cfg, scale=0.1)
    jconfig.set_root(str(tmp_path))
    from jlm_amd.model import LSTM_Model
    m = LSTM_Model(1).dev
    assert m.split_array is not None and m.mixed_idx == [] and m.ld_tm == 0 and m.mixed_calib is None


@pytest.mark.parametrize("mult,kept", [(1.0, True), (40.0, False), ((40.0, 1.0, 1.0), "first-split"), ("words<128", "head")])
def test_load_time_calibration_follows_the_logit_range(tmp_path, fake, monkeypatch, mult, kept):
    """DeviceModel._calibrate_mixed over the numpy double: the same model keeps its mixed rows with its output embeddings as they are and
    loses them with the embeddings x 40 (logits of +-40: the int8 cross terms would move path scores beyond the tolerance); with only
    the FIRST segment's embeddings x 40 -- the frequent words carry the mass, as in a trained model -- that segment goes to split rows and
    the other two stay mixed (round 5: jlm_vocab_lse_hybrid hosts the long split body); with every embedding x 10 and a bias that puts
    the mass on the first 128 words, those words alone (head_split) -- and the decode equals the oracle every time."""
    for k in ("JLM_MIXED_MAX_LSE_RMS", "JLM_MIXED_MAX_SPREAD", "JLM_LSE_MIXED"):
        monkeypatch.delenv(k, raising=False)
    root = str(tmp_path)
    cfg = synth.make_config(2000, 64, 200, "vtable", synth.wide_segs(2000))
    _lex, _rd = synth.write_lexicon(root, 2000, alphabet=12)
    w = synth.make_weights(cfg, scale=0.1)
    for key in list(w):
        if key.startswith("LM") and mult == "words<128":
            w[key] = (w[key] * np.float32(10.0)).astype(np.float32)
        elif key.startswith("LM"):
            f_ = mult[int(key[2:])] if isinstance(mult, tuple) else mult
            w[key] = (w[key] * np.float32(f_)).astype(np.float32)
    if kept == "head":
        w["b2"][:128] += np.float32(12.0)            # (embeddings x 10 everywhere, the mass on the first 128 words: 1.1e-5 rms on mixed rows, 7.5e-8 with that head on split rows)
        from jlm_amd.model import DeviceModel
        monkeypatch.setattr(DeviceModel, "HEAD_SPLITS", (128, 256))
    import json
    import pickle
    d = os.path.join(root, "train", "experiments", "1")
    os.makedirs(os.path.join(d, "weights"), exist_ok=True)
    with open(os.path.join(d, "config.json"), "wt") as f:
        f.write(json.dumps(cfg))
    with open(os.path.join(d, "weights", "lstm_weights.pkl"), "wb") as f:
        pickle.dump(w, f)
    jconfig.set_root(root)
    from jlm_amd.decoder import Decoder
    dec = Decoder(1)
    dec.perf_timing = False
    m = dec.model.dev
    assert m.mixed_calib is not None and m.mixed_calib["kept"] == bool(kept) and bool(m.mixed_idx) == bool(kept), m.mixed_calib
    if kept == "head":
        assert m.mixed_idx == [0, 1, 2] and m.mixed_head_split == [128, 0, 0] and m.mixed_calib["head_split"] == 128, m.mixed_calib
    if kept == "first-split":
        assert m.mixed_idx == [1, 2] and m.mixed_calib["split_segments"] == [0] and m.mixed_calib["lse_rms_diff_all_mixed"] > 1e-6, m.mixed_calib
    sents = synth.make_ragged_sentences(5, 2, 9, seed=5, alphabet=12)
    o = orc.OracleDecoder(root, 1)
    for s, g in zip(sents, dec.decode_batch(sents, beam_width=6)):
        want = o.decode(s, beam_width=6)
        assert [x for _, x in g] == [x for _, x in want]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in want], rtol=1e-6, atol=2e-5)


@pytest.mark.parametrize("seed", [0, 3, 5, 9])
def test_random_models_on_the_double(seed, tmp_path, fake):
    """A few of the GPU suite's random model draws (tests/random_models.py) through the host path on the numpy double."""
    from tests import random_models as rm
    rm.check(seed, str(tmp_path))


@pytest.mark.parametrize("name", ["wide-vtable", "wide128-tied"])
def test_calibration_on_decoded_paths(fx, fake, monkeypatch, name):
    """Round 6: Decoder.__init__ decodes a few synthetic sentences of its own lexicon and hands the kept hypotheses' word sequences to
    DeviceModel.calibrate_on_paths -- one more probe ("paths") beside the seeded word draws; the format decision stands on all of them."""
    f = fx(name)
    dec = _decoder(f, "static")
    m = dec.model.dev
    kinds = [p["kind"] for p in m.mixed_calib["probes"]]
    assert kinds.count("paths") == 1 and len(kinds) == len(m.CALIB_PROBES) + 1, kinds
    assert m.mixed_calib["kept"] and m.mixed_fmt == "mx6" and m.mixed_calib["lse_rms_diff"] == max(p["rms"] for p in m.mixed_calib["probes"])
    assert dec.perf_sen == 0 and dec.last_lattice is None            # the calibration decode leaves no trace in the decoder's counters
    # ... and the decode after it is the oracle's
    sents = synth.make_ragged_sentences(4, 2, 9, seed=5, alphabet=f["alphabet"])
    o = orc.OracleDecoder(f["root"], 1)
    for s, g in zip(sents, dec.decode_batch(sents, beam_width=8)):
        w = o.decode(s, beam_width=8)
        assert [x for _, x in g] == [x for _, x in w]
    monkeypatch.setenv("JLM_CALIB_PATHS", "0")
    dec0 = _decoder(f, "static")
    assert [p["kind"] for p in dec0.model.dev.mixed_calib["probes"]].count("paths") == 0
