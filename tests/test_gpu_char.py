"""GPU: the character-model decoder (jlm_amd/decoder_char.py; reference CharRNNDecoder, decoder/decoder.py:244-341, as evidently intended --
PARITY UNPINNED) on the HIP library against tests/golden/char.json (the reference's class with one method supplied at run time,
tools/make_golden.py gen_char) and against the CPU oracle on fresh seeded inputs.

Bar: identical 1-best, n-best order identical up to 1e-6 ties, scores within 2e-6 per character step + 2e-6 (a path of L kana takes at most
3 L + 1 steps on these fixtures: display strings of up to three characters).  The steps run on the kernels of LSTM_Model.predict (f32 state
rows, f32 logits, float32 softmax entries)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from jlm_amd import config as jconfig, synth      # noqa: E402
from oracle import jlm_oracle as orc              # noqa: E402
from tests import golden_cases as gc              # noqa: E402
from tests.test_gpu_decode import TIE_REL         # noqa: E402

pytestmark = pytest.mark.gpu

_DEC = {}


def _decoder(f):
    if f["root"] not in _DEC:
        jconfig.set_root(f["root"])
        from jlm_amd.decoder_char import CharRNNDecoder
        from jlm_amd import ops
        _DEC[f["root"]] = CharRNNDecoder(1)
        assert type(ops.backend()).__name__ == "HipOps"
    jconfig.set_root(f["root"])
    return _DEC[f["root"]]


def _bar(words_lists):
    steps = max(sum(len(w) for w in ws) for ws in words_lists) + 1
    return 2e-6 * steps + 2e-6


def _check(out, gold, tag, ordered=False):
    assert len(out) == len(gold), tag
    assert out[0][1] == gold[0][1], ("1-best differs", tag, out[0], gold[0])
    np.testing.assert_allclose([s for s, _ in out], [s for s, _ in gold], rtol=0, atol=_bar([w for _, w in gold]), err_msg=str(tag))
    if [w for _, w in out] == [w for _, w in gold]:
        return
    assert not ordered, ("generation order differs", tag)
    gscore = {tuple(w): s for s, w in gold}
    for i, (_s, w) in enumerate(out):
        if w != gold[i][1]:
            ref_mine = gscore.get(tuple(w), gold[-1][0])
            assert abs(ref_mine - gold[i][0]) <= TIE_REL * max(1.0, abs(gold[i][0])), ("n-best order differs beyond a tie", tag, i, w, gold[i])


@pytest.mark.parametrize("timed", [False, True], ids=["fast", "timed"])
@pytest.mark.parametrize("case", gc.CHAR_CASES, ids=[c[0] for c in gc.CHAR_CASES])
def test_decode_matches_the_wired_reference(case, timed, fx, golden_char):
    name, fixture, kwargs, spec = case
    f = fx(fixture)
    dec = _decoder(f)
    dec.perf_timing = timed
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_char[name]
    assert [g["input"] for g in gold] == sents
    n0 = len(dec.perf_log_lstm)
    outs = dec.decode_batch(sents, **kwargs)
    dec.perf_timing = False
    assert (len(dec.perf_log_lstm) > n0) == timed and len(dec.perf_log_lstm) == len(dec.perf_log_softmax)
    for si, out in enumerate(outs):
        _check(out, gold[si]["nbest"], (name, si), ordered=kwargs.get("beam_width", 10) is None)


def test_per_frame_beams_match_the_reference_traces(fx, golden_char):
    """every frame's kept paths (score, start frame, LAST character's index, number of nodes) of the first sentences"""
    name, fixture, kwargs, spec = gc.CHAR_CASES[0]
    f = fx(fixture)
    dec = _decoder(f)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_char[name]
    for si in range(gc.TRACE_SENTENCES):
        dec.decode(sents[si], **kwargs)
        frames = dec._last_frames[0]
        tr = gold[si]["trace"]
        assert len(frames) == len(tr)
        for i, (paths, gi) in enumerate(zip(frames, tr)):
            assert len(paths) == len(gi), (si, i)
            np.testing.assert_allclose([p.score for p in paths], [x[0] for x in gi], rtol=0, atol=2e-6 * (3 * i + 1) + 2e-6)
            got = sorted((p.start, p.idx, p.n_nodes) for p in paths)
            assert got == sorted(tuple(x[1:]) for x in gi), (si, i)


@pytest.mark.parametrize("fixture,n,lo,hi,beam", [("small-char", 48, 1, 24, 7), ("mid-char", 64, 20, 20, 10)])
def test_lockstep_batch_agrees_with_the_oracle_and_with_single_sentences(fixture, n, lo, hi, beam, fx):
    """fresh inputs: a ragged batch in lock step = the oracle sentence by sentence = decode() of one sentence at a time (scores to 1e-6: the
    rows of a batch are independent; the matrix kernels' tiling follows the row count)"""
    f = fx(fixture)
    dec = _decoder(f)
    o = orc.OracleCharRNNDecoder(f["root"], 1)
    sents = synth.make_ragged_sentences(n, lo, hi, seed=31, alphabet=f["alphabet"]) + [""]
    outs = dec.decode_batch(sents, beam_width=beam)
    assert outs[-1] == [(0.0, [])]
    for si in range(0, n, 7 if fixture.startswith("mid") else 1):
        _check(outs[si], o.decode(sents[si], beam_width=beam), (fixture, si))
    for si in (0, n // 2, n - 1):
        one = dec.decode(sents[si], beam_width=beam)
        assert one[0][1] == outs[si][0][1]
        np.testing.assert_allclose([s for s, _ in one], [s for s, _ in outs[si]], rtol=0, atol=1e-6 * max(1.0, abs(one[0][0])))
    # more sentences than one lock-step batch holds: the same lists
    dec.max_batch = 16
    try:
        again = dec.decode_batch(sents[:40], beam_width=beam)
    finally:
        dec.max_batch = 256
    for a, b in zip(again, outs[:40]):
        assert a[0][1] == b[0][1]
        np.testing.assert_allclose([s for s, _ in a], [s for s, _ in b], rtol=0, atol=1e-6 * max(1.0, abs(b[0][0]) if b else 1.0))


def test_eval_harness_on_a_character_model(fx, golden_char, monkeypatch, tmp_path):
    import contextlib, io, os
    name, fixture, argv = gc.CHAR_EVAL_CASE
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    monkeypatch.chdir(tmp_path)
    from jlm_amd import eval as jeval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        jeval.main(argv + ["--batch", "6"])
    gold = golden_char[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    with open(os.path.join("eval", gold["log_name"]), "r", encoding="utf-8") as fh:
        body = fh.read()
    assert body[:body.index("--- ")] == gold["log_body"]
