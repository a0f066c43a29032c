"""CPU: the n-gram baseline (jlm_amd/decoder_ngram.py, model_ngram.py) against golden vectors captured from the
reference's NGramDecoder / NGramModel / eval.py -ng (tests/golden/ngram.json, tools/make_golden.py --only ngram)."""
import contextlib
import io
import json
import os

import pytest

from jlm_amd import config as jconfig, synth
from tests import golden_cases as gc


@pytest.fixture(scope="module")
def golden_ngram():
    with open(os.path.join(os.path.dirname(__file__), "golden", "ngram.json"), "r", encoding="utf-8") as f:
        return json.load(f)


@pytest.mark.parametrize("case", gc.NGRAM_CASES, ids=[c[0] for c in gc.NGRAM_CASES])
def test_ngram_decoder_matches_reference(case, fx, golden_ngram):
    name, fixture, order, kwargs, spec = case
    f = fx(fixture)
    jconfig.set_root(f["root"])
    from jlm_amd.decoder_ngram import NGramDecoder
    with contextlib.redirect_stdout(io.StringIO()):
        dec = NGramDecoder(1, ngram_order=order)
    sents = gc.ngram_sentences(spec, f["alphabet"], name == gc.NGRAM_CASES[0][0])
    gold = golden_ngram[name]
    assert [g["input"] for g in gold] == sents
    n_empty = 0
    for s, g in zip(sents, gold):
        got = dec.decode(s, **kwargs)
        assert [[a, list(b)] for a, b in got] == g["nbest"], (name, s)      # same float operations: equal to the last bit
        if got:
            assert dec.model.evaluate(list(got[0][1])) == g["evaluate_best"]
        else:
            n_empty += 1
    assert (n_empty > 0) == (name == gc.NGRAM_CASES[0][0])                   # the uncovered input: no path, not an <unk> path
    assert dec.perf_sen == len(sents) and len(dec.perf_log) > 0


def test_ngram_model_lookup_rules(fx):
    f = fx("small-tied")
    jconfig.set_root(f["root"])
    from jlm_amd.model_ngram import NGramModel, UNSEEN_COST
    with contextlib.redirect_stdout(io.StringIO()):
        m = NGramModel(ngram_file='lm3', ngram_order=3)
    assert ('<eos>',) in m.model and m.model[('<eos>',)][1] is not None      # "<s>" came last: its back-off is kept
    assert m.predict(['no such word']) == UNSEEN_COST
    some = next(k for k in m.model if len(k) == 3)
    assert m.predict(list(some)) == m.model[some][0]
    assert m.predict(['x', 'y'] + list(some)) == m.model[some][0]            # only the last `order` words count
    assert m.predict(['no such word', some[2]]) == m.predict([some[2]])      # unseen history: shorter suffix


def test_eval_harness_with_the_ngram_decoder(fx, golden_ngram, monkeypatch, tmp_path):
    name, fixture, argv = gc.NGRAM_EVAL_CASE
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    monkeypatch.chdir(tmp_path)
    from jlm_amd import eval as jeval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        jeval.main(argv)
    gold = golden_ngram[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    with open(os.path.join("eval", gold["log_name"]), "r", encoding="utf-8") as fh:
        body = fh.read()
    assert body[:body.index("--- ")] == gold["log_body"]


@pytest.mark.skipif(not os.path.exists("/root/reference/decoder/eval.py"), reason="the reference is only present in the build container")
def test_reference_eval_py_with_ng_runs_unchanged_over_compat(fx, golden_ngram, monkeypatch, tmp_path):
    """The reference's own eval.py -ng True, its NGramDecoder import resolved by compat/decoder_ngram.py."""
    name, fixture, argv = gc.NGRAM_EVAL_CASE
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    monkeypatch.chdir(tmp_path)
    os.makedirs("eval")
    from tools import run_reference_eval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        run_reference_eval.run("/root/reference/decoder/eval.py", argv)
    gold = golden_ngram[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    with open(os.path.join("eval", gold["log_name"]), "r", encoding="utf-8") as fh:
        body = fh.read()
    assert body[:body.index("--- ")] == gold["log_body"]
