"""CPU: the character-model decoder (reference CharRNNDecoder, decoder/decoder.py:244-341, as evidently intended -- PARITY UNPINNED,
DESIGN.md 8: the reference's class cannot run as shipped; tests/golden/char.json holds the output of the reference class with ONE
method supplied at run time, tools/make_golden.py gen_char).

  * the oracle's restatement (oracle/jlm_oracle.py build_char_lattice / char_decode) against those vectors: 1e-10;
  * CharVocab against the index the fixture generator computes from the reference's rule (train/data.py:28-40);
  * the product's host logic (jlm_amd/decoder_char.py: lattice, string dedup, lock-step batching, state pool, gathers) end to end on the
    numpy double of the C ABI (tests/fake_hip.py) against the same vectors.  The kernels: tests/test_gpu_char.py."""
import numpy as np
import pytest

from jlm_amd import config as jconfig, synth
from oracle import jlm_oracle as orc
from tests import fake_hip
from tests import golden_cases as gc

SMALL = [c for c in gc.CHAR_CASES if c[0].startswith("small-")]


@pytest.fixture()
def fake(monkeypatch):
    return fake_hip.install(monkeypatch)


@pytest.mark.parametrize("case", gc.CHAR_CASES, ids=[c[0] for c in gc.CHAR_CASES])
def test_oracle_matches_the_wired_reference(case, fx, golden_char):
    name, fixture, kwargs, spec = case
    f = fx(fixture)
    dec = orc.OracleCharRNNDecoder(f["root"], 1)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_char[name]
    assert [g["input"] for g in gold] == sents
    for si, s in enumerate(sents[:6 if fixture.startswith("mid") else None]):
        out = dec.decode(s, **kwargs)
        g = gold[si]["nbest"]
        assert [w for _, w in out] == [w for _, w in g], (name, si)
        np.testing.assert_allclose([sc for sc, _ in out], [sc for sc, _ in g], rtol=1e-10, atol=1e-10)
        if "trace" in gold[si]:
            tr = dec.last_trace
            assert len(tr) == len(gold[si]["trace"])
            for (scores, tail), gi in zip(tr, gold[si]["trace"]):
                np.testing.assert_allclose(scores, [x[0] for x in gi], rtol=1e-10, atol=1e-10)
                assert [list(t) for t in tail] == [x[1:] for x in gi]


def test_char_vocab(fx):
    f = fx("small-char")
    jconfig.set_root(f["root"])
    from jlm_amd.data import CharVocab
    v = CharVocab(f["cfg"]["vocab_size"])
    assert v.c2i == synth.char_index(f["lexicon"], f["cfg"]["vocab_size"])
    assert v.c2i["<unk>"] == 0 and v.c2i["<eos>"] == 1 and len(v) == len(v.c2i) and v.i2c[1] == "<eos>"
    assert 0 < len(v.w2i) <= f["cfg"]["vocab_size"]        # the word index is still there (eval.py:35-37 reads it; equal entries collapse)
    # the model's softmax has one row per character
    import pickle, os
    with open(os.path.join(f["root"], "train", "experiments", "1", "weights", "lstm_weights.pkl"), "rb") as fh:
        assert pickle.load(fh)["b2"].shape[0] == len(v)


def test_lattice_of_the_character_decoder(fx, fake):
    """one node per distinct display string of a (start, reading), first-character index, <unk> fallback -- against the oracle's lattice"""
    f = fx("small-char")
    jconfig.set_root(f["root"])
    from jlm_amd.decoder_char import CharRNNDecoder
    dec = CharRNNDecoder(1)
    o = orc.OracleCharRNNDecoder(f["root"], 1)
    dups = 0
    for s in synth.make_ragged_sentences(40, 1, 14, seed=21, alphabet=f["alphabet"]) + ["ンン"]:      # (a kana outside the alphabet: <unk> nodes)
        ends = orc.build_char_lattice(s, o.full_lexicon, o.full_reading_dict, o.words, o.w2i)
        assert dec._ends(s) == ends
        bl = dec._build_lattice(s)
        assert [[(n.start_idx, n.reading_length, n.word_idx, n.word) for n in bl[i]] for i in range(len(s) + 1)] == ends
        for i in range(len(s)):
            for j in range(len(s) - i):
                ids = o.full_reading_dict.get(s[i:i + j + 1], [])
                disp = [o.full_lexicon[k][0].split("/")[0] for k in ids if o.full_lexicon[k][0] in o.words]
                dups += len(disp) - len(set(disp))
    assert dups > 0          # the dedup of decoder.py:116-122 was exercised
    assert dec._build_lattice("ン")[1][0].word_idx == 0


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_decoder_matches_the_wired_reference(case, fx, fake, golden_char):
    name, fixture, kwargs, spec = case
    f = fx(fixture)
    jconfig.set_root(f["root"])
    from jlm_amd.decoder_char import CharRNNDecoder
    dec = CharRNNDecoder(1)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_char[name]
    outs = dec.decode_batch(sents, **kwargs)              # ragged batch, frames in lock step
    for si, out in enumerate(outs):
        g = gold[si]["nbest"]
        assert [w for _, w in out] == [w for _, w in g], (name, si)
        np.testing.assert_allclose([sc for sc, _ in out], [sc for sc, _ in g], rtol=1e-5, atol=1e-4)
    # one sentence at a time: the same lists
    for si in (0, len(sents) - 1):
        one = dec.decode(sents[si], **kwargs)
        assert [w for _, w in one] == [w for _, w in outs[si]]
        np.testing.assert_allclose([sc for sc, _ in one], [sc for sc, _ in outs[si]], rtol=0, atol=1e-9)
    assert dec.backward_lookup[0][0].word == "<eos>" and dec.perf_sen == len(sents) + 2
    if kwargs.get("vocab_select"):
        assert dec.lattice_vocab == sorted(set(dec.lattice_vocab)) and set(range(20)) <= set(dec.lattice_vocab)


def test_edges(fx, fake):
    f = fx("small-char")
    jconfig.set_root(f["root"])
    from jlm_amd import decoder as jdec
    from jlm_amd.decoder_char import CharRNNDecoder
    assert jdec.CharRNNDecoder is CharRNNDecoder         # ``from decoder import Decoder, CharRNNDecoder`` (eval.py:7)
    dec = CharRNNDecoder(1)
    assert dec.decode("") == [(0.0, [])]
    assert dec.decode_batch([]) == []
    mixed = dec.decode_batch(["", "アイ", ""], beam_width=4)
    assert mixed[0] == [(0.0, [])] and mixed[2] == [(0.0, [])] and len(mixed[1]) >= 1
    with pytest.raises(ValueError):
        dec.decode("ア", beam_width=0)
    dec.max_unpruned_paths = 5
    with pytest.raises(ValueError, match="max_unpruned_paths"):
        dec.decode("アイウエオカ", beam_width=None)
    # per-step timings (eval.py:104-121 reads the lists): one entry per batch of steps
    dec.perf_timing = True
    n0 = len(dec.perf_log_lstm)
    dec.decode("アイウ", beam_width=3)
    assert len(dec.perf_log_lstm) > n0 and len(dec.perf_log_lstm) == len(dec.perf_log_softmax)
    # a character model is refused by the word decoders (their lattice indexes a softmax over words)
    with pytest.raises(ValueError, match="character model"):
        jdec.Decoder(1)


def _eval_body(path):
    with open(path, "r", encoding="utf-8") as fh:
        body = fh.read()
    return body[:body.index("--- ")] if "--- " in body else body


@pytest.mark.parametrize("batch", [1, 8])
def test_eval_harness_on_a_character_model(batch, fx, fake, golden_char, monkeypatch, tmp_path):
    """jlm_amd.eval picks CharVocab + CharRNNDecoder by config['char_rnn'] (eval.py:35-36, 43-44): log body and hit counts of the reference's
    eval.py on the same fixture"""
    import contextlib, io, os
    name, fixture, argv = gc.CHAR_EVAL_CASE
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    monkeypatch.chdir(tmp_path)
    from jlm_amd import eval as jeval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        jeval.main(argv + ["--batch", str(batch)])
    gold = golden_char[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    assert _eval_body(os.path.join("eval", gold["log_name"])) == gold["log_body"]


@pytest.mark.skipif(not __import__("os").path.exists("/root/reference/decoder/eval.py"), reason="the reference is only present in the build container")
def test_reference_eval_py_runs_unchanged_on_a_character_model(fx, fake, golden_char, monkeypatch, tmp_path):
    """the reference's eval.py, unchanged, over compat/ (``from decoder import Decoder, CharRNNDecoder``, ``from train.data import Vocab,
    CharVocab``): the class it gets is jlm_amd's"""
    import contextlib, io, os
    name, fixture, argv = gc.CHAR_EVAL_CASE
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    monkeypatch.chdir(tmp_path)
    os.makedirs("eval")
    from tools import run_reference_eval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        run_reference_eval.run("/root/reference/decoder/eval.py", argv)
    gold = golden_char[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    assert _eval_body(os.path.join("eval", gold["log_name"])) == gold["log_body"]
