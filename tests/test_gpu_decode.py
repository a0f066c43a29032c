"""GPU: end-to-end parity of the reference-compatible classes, running on the
HIP library, against (a) the golden vectors captured from the reference and
(b) the CPU oracle on the same seeded inputs; plus size-independent properties
at BASELINE.json's full batch sizes.

Bars (BASELINE.json north_star): step logits <= 1e-4 relative, identical 1-best
decode strings.  Path scores are sums of ~20 float32-derived terms of size ~10,
so they are compared at 1e-5 relative / 1e-3 absolute."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from jlm_amd import config as jconfig, synth      # noqa: E402
from oracle import jlm_oracle as orc              # noqa: E402
from tests import golden_cases as gc              # noqa: E402

pytestmark = pytest.mark.gpu

_DEC = {}


def _decoder(f, kind):
    key = (f["root"], kind)
    if key not in _DEC:
        jconfig.set_root(f["root"])
        from jlm_amd.decoder import Decoder
        from jlm_amd.decoder_dynamic import DynamicDecoder
        _DEC[key] = (DynamicDecoder if kind == "dynamic" else Decoder)(1)
    jconfig.set_root(f["root"])
    return _DEC[key]


TIE_REL = 1e-6        # two hypotheses whose reference scores are this close (relative) may come out in either order


# (JLM_PRECISION=f32 -- plain f32 accumulation on the f32 matrix pipe, a supported knob -- holds 1.5e-6 per frame: 2.4e-5 measured on
#  peaked20-tied/dynamic at L = 20 where the default split-f16 path holds 1.1e-5; the knob sweep, tools/gpu_knobs.sh, runs this file under it)
SCORE_ATOL_PER_FRAME = 1.5e-6 if os.environ.get("JLM_PRECISION", "f16x3") == "f32" else 1e-6
SCORE_ATOL_FLOOR = 2e-6


SCORE_ATOL_FLAT = 2e-5            # rounds 1-4's flat bar: kept as a cap for inputs of up to 20 kana (round-5 advice: the per-frame formula
                                  # gives 2.3e-5 at L = 20 -- looser than what those cases were held to before)


def score_atol(n_kana):
    """The score bar scales with the path length: a path score is a sum over the L + 1 frames of (log-normaliser - edge logit)
    terms of size ~5..10, each good to ~1e-6 absolute on the f32-grade matrix products (north_star's own bars are 1e-4 relative
    on the step logits and identical 1-best strings).  1e-6 per frame + 2e-6: 2.3e-5 at the headline L = 20 (rounds 1-4 used a
    flat 2e-5), 1.3e-5 at L = 10, 4.3e-5 at L = 40 -- 2e-7 of the scores themselves (~50 / 100 / 230)."""
    bar = SCORE_ATOL_PER_FRAME * (n_kana + 1) + SCORE_ATOL_FLOOR
    # (the flat cap is the DEFAULT path's: under JLM_PRECISION=f32 the f32 pipe measures 2.0e-5 .. 2.4e-5 on peaked20-tied/dynamic at L = 20)
    capped = n_kana <= 20 and os.environ.get("JLM_PRECISION", "f16x3") != "f32"
    return min(bar, SCORE_ATOL_FLAT) if capped else bar


def _check_nbest(out, gold, tag, n_kana=20):
    """1-best identical, scores within score_atol(sentence length) and the n-best ORDER identical -- except between
    hypotheses whose reference scores are within TIE_REL of each other (float32-derived path scores cannot order those)."""
    assert len(out) == len(gold), tag
    assert out[0][1] == gold[0][1], ("1-best differs", tag, out[0], gold[0])
    np.testing.assert_allclose([s for s, _ in out], [s for s, _ in gold], rtol=0, atol=score_atol(n_kana), err_msg=str(tag))
    if [w for _, w in out] == [w for _, w in gold]:
        return True
    gscore = {tuple(w): s for s, w in gold}
    for i, (_s, w) in enumerate(out):
        if w == gold[i][1]:
            continue
        ref_here = gold[i][0]
        # the hypothesis at this rank must be one the reference scores within the tie distance of its own rank-i score
        # (a hypothesis that fell off the reference's list altogether can only have tied with its last entry)
        ref_mine = gscore.get(tuple(w), gold[-1][0])
        assert abs(ref_mine - ref_here) <= TIE_REL * max(1.0, abs(ref_here)), ("n-best order differs beyond a tie", tag, i, w, gold[i])
    return False


@pytest.mark.parametrize("name", gc.LM_FIXTURES)
def test_step_logits_within_1e4_of_reference(name, fx, golden_lm):
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
            ysel = y if kind == "subset" else y[:, cols]
            psel = pred if kind == "subset" else pred[:, cols]
            yref = golden_lm[key + "/y"]
            # <= 1e-4 relative on the step logits (relative to the row's logit scale; element-wise
            # relative error is meaningless for logits that happen to be ~0)
            scale = np.abs(yref).max(axis=1, keepdims=True)
            rel = np.abs(ysel - yref) / scale
            assert rel.max() <= 1e-4, (key, rel.max())
            # ... and element-wise: |dy| <= 1e-4 max(|y|, 1)
            el = np.abs(ysel - yref) / np.maximum(np.abs(yref), 1.0)
            assert el.max() <= 1e-4, (key, el.max())
            np.testing.assert_allclose(h, golden_lm[key + "/h"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(c, golden_lm[key + "/c"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(psel, golden_lm[key + "/pred"], rtol=2e-4)
            np.testing.assert_allclose(pred.sum(axis=1), golden_lm[key + "/predsum"], rtol=1e-4)


@pytest.mark.parametrize("fast", [False, True], ids=["timed", "fast"])
@pytest.mark.parametrize("case", gc.DECODE_CASES, ids=[c[0] for c in gc.DECODE_CASES])
def test_decode_matches_reference_golden(case, fast, fx, golden_decode):
    """timed: per-frame HIP events, one stream (what eval.py's perf logs need); fast: the throughput path --
    alternating streams, side stream, graph replay when enabled"""
    name, fixture, kind, kwargs, spec = case
    f = fx(fixture)
    dec = _decoder(f, kind)
    dec.perf_timing = not fast
    dec.compat_quirks = gc.is_quirk_case(name)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_decode[name]
    assert [g["input"] for g in gold] == sents
    if kwargs.get("random_sampling"):
        outs = []
        for si, s in enumerate(sents):
            np.random.seed(gc.RANDOM_SAMPLING_SEED + si)
            outs.append(dec.decode(s, **kwargs))
    else:
        outs = dec.decode_batch(sents, **kwargs)
    dec.compat_quirks = False
    for si, out in enumerate(outs):
        _check_nbest(out, gold[si]["nbest"], (name, si), len(sents[si]))       # order identical unless two reference scores tie to 1e-6


@pytest.mark.parametrize("fixture", ["small-vtable", "small-tied"])
def test_f32_pipe_path_agrees_with_split_path(fixture, fx, monkeypatch):
    """JLM_PRECISION=f32 keeps the whole decode on the f32 matrix pipe (the round's first
    kernels); it must give the same n-best as the default split-f16 decode."""
    if os.environ.get("JLM_PRECISION", "f16x3") != "f16x3":
        pytest.skip("the suite is running with JLM_PRECISION=f32: there is no split path to compare with")
    f = fx(fixture)
    jconfig.set_root(f["root"])
    from jlm_amd.decoder import Decoder
    sents = synth.make_ragged_sentences(12, 2, 16, seed=77, alphabet=f["alphabet"])
    split = _decoder(f, "static")
    assert split.model.dev.split_array is not None and split.model.dev.split_lstm
    monkeypatch.setenv("JLM_PRECISION", "f32")
    plain = Decoder(1)
    assert plain.model.dev.split_array is None and not plain.model.dev.split_lstm
    a = split.decode_batch(sents, beam_width=7)
    b = plain.decode_batch(sents, beam_width=7)
    for x, y in zip(a, b):
        assert [w for _, w in x] == [w for _, w in y]
        np.testing.assert_allclose([v for v, _ in x], [v for v, _ in y], rtol=0, atol=2e-5)


@pytest.mark.parametrize("fixture,n_sent,beam", [("wide-vtable", 12, 7), ("wide-dsoftmax", 12, 7), ("wideh-vtable", 12, 7), ("wide128-tied", 12, 7),
                                                 ("mid-vtable", 64, 10), ("mid-tied", 64, 10)])
def test_mixed_rows_normaliser_agrees_with_split_rows(fixture, n_sent, beam, fx, monkeypatch):
    """Segments of width 200 / 100 / 50 run the full-vocabulary normaliser on mixed rows (f16 hi.hi + int8 cross terms,
    jlm_pack_t_mixed + jlm_vocab_lse_mixed, include/jlm_hip.h ABI 7) by default; JLM_LSE_MIXED=0 keeps every segment on its
    split rows.  Same n-best, scores within the decode tolerance; the small fixtures also against the oracle."""
    if os.environ.get("JLM_PRECISION", "f16x3") != "f16x3" or os.environ.get("JLM_LSE_MIXED", "1") == "0":
        pytest.skip("the suite is running without the mixed rows")
    f = fx(fixture)
    jconfig.set_root(f["root"])
    from jlm_amd.decoder import Decoder
    sents = synth.make_ragged_sentences(n_sent, 2, 16, seed=78, alphabet=f["alphabet"])
    mixed = _decoder(f, "static")
    if fixture.startswith("wideh"):          # the short last segment stays on split rows: jlm_vocab_lse_hybrid
        assert mixed.model.dev.mixed_idx == [0, 1] and mixed.model.dev.ld_tm == 360
    elif fixture.endswith("-tied"):          # a contraction that fills its last block: biases outside the rows (k = 128, 256)
        assert mixed.model.dev.mixed_idx == [0] and mixed.model.dev.b2_log2 is not None
    else:
        assert mixed.model.dev.mixed_idx == [0, 1, 2] and mixed.model.dev.ld_tm == 424
    a = mixed.decode_batch(sents, beam_width=beam)
    monkeypatch.setenv("JLM_LSE_MIXED", "0")
    split = Decoder(1)
    assert split.model.dev.mixed_idx == []
    b = split.decode_batch(sents, beam_width=beam)
    for x, y in zip(a, b):
        assert [w for _, w in x] == [w for _, w in y]
        np.testing.assert_allclose([v for v, _ in x], [v for v, _ in y], rtol=0, atol=2e-5)
    if fixture.startswith("wide"):
        from oracle import jlm_oracle as orc
        o = orc.OracleDecoder(f["root"], 1)
        for s, x in zip(sents, a):
            _check_nbest(x, o.decode(s, beam_width=beam), (fixture, s))


@pytest.mark.parametrize("fixture,kind,kw", [
    ("small-vtable", "static", {}), ("small-tied", "static", {"vocab_select": True}),
    ("small-tied", "dynamic", {"vocab_select": True}), ("small-untied", "static", {}),
    ("small-tied-sn", "static", {})])
def test_timed_frame_loop_equals_fast_frame_loop(fixture, kind, kw, fx):
    """torch.ops.jlm.decode_frames (one op per batch) in its two forms -- fast: alternating streams, edge logits on a side
    stream; timed: one stream, HIP events around every frame's kernel groups (what eval.py's perf logs and bench.py's
    per-kernel durations read) -- launches the same kernels on the same operands: identical n-best, bit-identical scores.
    The untied model (k = H > 256: tile-form normaliser) runs inside the same op."""
    f = fx(fixture)
    dec = _decoder(f, kind)
    eng = dec._engine
    sents = synth.make_ragged_sentences(14, 1, 17, seed=91, alphabet=f["alphabet"])
    dec.perf_timing = False
    a = dec.decode_batch(sents, beam_width=6, **kw)
    dec.perf_timing = True
    n0 = len(dec.perf_log_lstm)
    try:
        b = dec.decode_batch(sents, beam_width=6, **kw)
    finally:
        dec.perf_timing = False
    steps = max(len(s) for s in sents)
    assert len(dec.perf_log_lstm) - n0 == steps and all(t > 0 for t in dec.perf_log_lstm[n0:])
    assert len(eng.last_kernel_ms["gate_gemm"]) == steps and len(eng.last_fix_timing) == steps + 1
    for x, y in zip(a, b):
        assert [w for _, w in x] == [w for _, w in y]
        assert [v for v, _ in x] == [v for v, _ in y]


@pytest.mark.parametrize("fixture,kind,kw", [("small-vtable", "static", {}), ("small-tied", "static", {"vocab_select": True}),
                                             ("small-tied", "dynamic", {"vocab_select": True}),
                                             ("wide-vtable", "static", {})])         # mixed rows (Tm of the plans reused in flight)
def test_pipelined_chunks_equal_serial_decode(fixture, kind, kw, fx):
    """Race hunt (tools/probes/soak_race.py at full size): 24 ragged chunks through the pipelined path -- one batch in flight per
    launch stream, page-locked lattice blocks handed back to the pool and reused, frame-loop op, lattice prefetch threads, plans reused while others are in flight -- against one chunk at
    a time on one stream, timed (no side stream either).  Same kernels, same operands: bit-identical results."""
    f = fx(fixture)
    dec = _decoder(f, kind)
    eng = dec._engine
    dec.perf_timing = False
    sents = synth.make_ragged_sentences(24 * 48, 1, 22, seed=123, alphabet=f["alphabet"])
    keep = (dec.max_batch, eng.n_streams, dec.perf_timing, dec.pipeline_depth, dec.prefetch_workers)
    share = eng.lse_share_pct
    try:
        dec.max_batch, dec.prefetch_workers = 48, 3
        # (the engine's default CU share is 100 %: the normaliser's column cuts -- the summation order -- do not depend on what else is
        #  in flight or on how many chunks the call has; a share below 100 would make them)
        assert eng.lse_share_pct in (0, 100)
        fast = dec.decode_batch(sents, beam_width=8, **kw)
        chunks = dec._chunks(sents, 8)                     # the same device batches (dealt by decreasing length), one at a time
        eng.n_streams, dec.perf_timing, dec.pipeline_depth, dec.prefetch_workers = 1, True, 0, 1
        slow = [None] * len(sents)
        for idx in chunks:
            for j, r in zip(idx, dec.decode_batch([sents[j] for j in idx], beam_width=8, **kw)):
                slow[j] = r
    finally:
        dec.max_batch, eng.n_streams, dec.perf_timing, dec.pipeline_depth, dec.prefetch_workers = keep
        eng.lse_share_pct = share
    assert len(fast) == len(slow) == len(sents)
    assert fast == slow


def test_single_sentence_equals_batch(fx):
    f = fx("small-vtable")
    dec = _decoder(f, "static")
    dec.perf_timing = True
    sents = synth.make_ragged_sentences(9, 1, 18, seed=31, alphabet=f["alphabet"])
    batch = dec.decode_batch(sents, beam_width=6)
    for s, b in zip(sents, batch):
        one = dec.decode(s, beam_width=6)
        assert [w for _, w in one] == [w for _, w in b]
        np.testing.assert_allclose([x for x, _ in one], [x for x, _ in b], rtol=0, atol=1e-9)
    assert dec.perf_log_lstm and dec.perf_log_softmax and all(t >= 0 for t in dec.perf_log_lstm)


def _readings_ok(words, text):
    r = "".join(w.split("/")[1] if "/" in w else w for w in words)
    return r == text


N_ORACLE = 32          # sentences of every full-size case that are decoded by the oracle as well


@pytest.mark.parametrize("fixture,kind,kwargs,n,beam", [
    ("mid-vtable", "static", {}, 256, 10),                                  # BASELINE.json configs[1]
    ("mid-tied", "dynamic", dict(vocab_select=True), 256, 10),              # configs[3]
    ("big-tied", "static", {}, 1024, 20),                                   # configs[2]
    ("mid-tied", "static", {}, 1024, 10),                                   # configs[4]: one GPU's share of the 8 192 sentences
])
def test_full_size_properties(fixture, kind, kwargs, n, beam, fx):
    """Size-independent properties at the benchmark's batch sizes + oracle parity on a sample."""
    f = fx(fixture)
    dec = _decoder(f, kind)
    dec.perf_timing = False
    sents = synth.make_sentences(n, 20, seed=2024, alphabet=f["alphabet"])
    out = dec.decode_batch(sents, beam_width=beam, **kwargs)
    assert len(out) == n
    for s, nb in zip(sents, out):
        assert 1 <= len(nb) <= min(10, beam)
        sc = [x for x, _ in nb]
        assert all(np.isfinite(sc)) and sc == sorted(sc)                   # ascending -log p
        for _, words in nb:
            assert _readings_ok(words, s), (s, words)                      # every path spells the input
        assert len({tuple(w) for _, w in nb}) == len(nb)                   # hypotheses are distinct
    again = dec.decode_batch(sents, beam_width=beam, **kwargs)             # idempotent / deterministic
    assert [[w for _, w in nb] for nb in again] == [[w for _, w in nb] for nb in out]
    # batch-composition independence: a sub-batch gives the same answers
    sub = dec.decode_batch(sents[5:37], beam_width=beam, **kwargs)
    # (the vocabulary ranges of the LSE kernel are sized from the batch's row count, so float32
    # summation order -- hence the last bits of a score -- may differ between batch sizes)
    for a, b in zip(sub, out[5:37]):
        assert [w for _, w in a][0] == [w for _, w in b][0]
        np.testing.assert_allclose([x for x, _ in a], [x for x, _ in b], rtol=0, atol=2e-5)
    # oracle on a sample of the same inputs, spread over the batch (first, last and evenly in between)
    o = (orc.OracleDynamicDecoder if kind == "dynamic" else orc.OracleDecoder)(f["root"], 1)
    for si in sorted({int(round(x)) for x in np.linspace(0, n - 1, N_ORACLE)}):
        want = o.decode(sents[si], beam_width=beam, **kwargs)
        _check_nbest(out[si], want, (fixture, si))


@pytest.mark.parametrize("case", [c for c in gc.DECODE_CASES if c[0] in (
    "small-tied/static", "small-vtable/static-vs", "small-tied/dynamic", "small-tied-sn/dynamic", "mid-tied/static",
    "mid-tied/dynamic", "mid-vtable/static")], ids=lambda c: c[0])
def test_per_frame_beams_match_reference_traces(case, fx, golden_decode):
    """Not only the final n-best: every frame's surviving hypotheses (score, last node's start
    frame and softmax row, path length) equal what the reference's Path objects held."""
    name, fixture, kind, kwargs, spec = case
    f = fx(fixture)
    dec = _decoder(f, kind)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_decode[name]
    n = sum(1 for g in gold if "trace" in g)
    dec.decode_batch(sents[:n], **kwargs)
    p = dec._engine.last_state
    lat = dec.last_lattice
    B, beam, rmax = lat.n_sent, lat.beam, lat.n_sent * lat.beam
    score, bp, node, cnt = (getattr(p, k).cpu().numpy() for k in ("score", "bp", "node", "cnt"))
    for s in range(n):
        trace = gold[s]["trace"]
        assert len(trace) == len(sents[s]) + 1
        for fr, want in enumerate(trace):
            k = int(cnt[fr * B + s])
            assert k == len(want), (name, s, fr)
            g0 = fr * rmax + s * beam
            for r, (w_score, w_start, w_word, w_len) in enumerate(want):
                g = g0 + r
                nd = int(node[g])
                assert int(lat.node_start[nd]) == w_start and int(lat.node_word[nd]) == w_word, (name, s, fr, r)
                depth, q = 0, g
                while q >= 0:
                    depth, q = depth + 1, int(bp[q])
                assert depth == w_len
                assert abs(score[g] - w_score) <= 1e-5 * abs(w_score) + 1e-3, (name, s, fr, r, score[g], w_score)
