"""Generate golden vectors by running the REFERENCE (imported unmodified from
/root/reference) on the seeded synthetic fixtures.

Runs only in the build container (the reference never travels to the GPU box).
Writes data only -- inputs regenerate from seeds (tests/golden_cases.py,
jlm_amd/synth.py), expected outputs go to tests/golden/:

  lm_steps.npz      LSTM_Model.predict_with_context outputs
  decode.json       Decoder.decode / DynamicDecoder.decode n-best lists (+ per
                    frame beams, read from the reference's Path objects)
  eval.json         decoder/eval.py run unchanged via runpy: hit counts and log
  ngram.json        NGramDecoder.decode n-best lists, NGramModel.evaluate, eval.py -ng
  char.json         CharRNNDecoder.decode n-best lists (+ per frame beams): the reference's class with ONE method supplied
                    at run time (it cannot run as shipped; gen_char below)

Usage:  python tools/make_golden.py [--only lm|decode|eval|ngram|char] [--filter substr]
"""
import argparse
import contextlib
import io
import json
import os
import runpy
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"

from jlm_amd import synth                       # noqa: E402
from tests import golden_cases as gc            # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
_roots = {}
_alpha = {}


def fixture_root(name):
    if name not in _roots:
        d = os.path.join(tempfile.gettempdir(), "jlm_golden_fx", name)
        os.makedirs(d, exist_ok=True)
        _cfg, _lex, _rd, alphabet = synth.build_fixture(d, name)
        _roots[name], _alpha[name] = d, alphabet
    return _roots[name]


def import_reference(root):
    """SURVEY.md 8(c) recipe: decoder/ first on sys.path, patch config before
    the other modules import from it."""
    for m in ("config", "model", "decoder", "decoder_dynamic", "decoder_ngram", "train", "train.data", "japanese",
              "model_ngram"):
        sys.modules.pop(m, None)
    for p in (REF, os.path.join(REF, "decoder")):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "decoder"))
    with contextlib.redirect_stdout(io.StringIO()):
        import config
        config.root_path = root
        config.data_path = os.path.join(root, "data")
        config.train_path = os.path.join(root, "train")
        config.experiment_path = os.path.join(root, "train", "experiments")
        import model
        import decoder
        import decoder_dynamic
    return model, decoder, decoder_dynamic


def gen_lm():
    out = {}
    for fx in gc.LM_FIXTURES:
        root = fixture_root(fx)
        model, _, _ = import_reference(root)
        with contextlib.redirect_stdout(io.StringIO()):
            lm = model.LSTM_Model(1)
        cfg = lm.config
        for rows in gc.LM_ROWS:
            idx, subset, cols, h0, c0 = gc.lm_inputs(cfg, rows)
            for kind in ("full", "subset"):
                if kind == "subset" and not cfg["share_embedding"]:
                    continue          # reference model.py:189 indexes rows of UM[H,V]: IndexError
                vocab = subset if kind == "subset" else None
                h, c = h0.copy(), c0.copy()
                for step in range(gc.LM_STEPS):
                    (pred, y, _t1, _t2), h, c = lm.predict_with_context(idx[step], h, c, vocab)
                key = "%s/%s/R%d" % (fx, kind, rows)
                out[key + "/h"] = h
                out[key + "/c"] = c
                ysel = y if kind == "subset" else y[:, cols]
                psel = pred if kind == "subset" else pred[:, cols]
                out[key + "/y"] = ysel
                out[key + "/pred"] = psel
                out[key + "/ymax"] = np.amax(y, axis=1)
                m = np.amax(y, axis=1, keepdims=True)
                out[key + "/lse"] = (m + np.log(np.sum(np.exp(y - m), axis=1, keepdims=True)))[:, 0]
                out[key + "/predsum"] = np.sum(pred, axis=1)
                print("lm", key, y.shape)
    np.savez_compressed(os.path.join(GOLD, "lm_steps.npz"), **out)


def _snap(paths):
    """(score, start_idx of last node, word_idx of last node, #nodes) per path."""
    return [[float(p.neg_log_prob), int(p.nodes[-1].start_idx), int(p.nodes[-1].word_idx), len(p.nodes)] for p in paths]


def _install_trace(dec, kind):
    """Wrap the reference's _build_current_frame (the frame dict is one of its
    arguments) so the per-frame beams can be read without touching the source."""
    store = {"frame": None, "snaps": {}}
    orig = dec._build_current_frame

    def wrapped(frame, i, *a, **k):
        store["frame"] = frame
        r = orig(frame, i, *a, **k)
        if kind == "dynamic":
            store["snaps"][i] = _snap(frame[i])      # pruned inside; later mutated in place
        return r

    dec._build_current_frame = wrapped
    return store


def gen_decode(flt=None):
    path = os.path.join(GOLD, "decode.json")
    results = {}
    if flt and os.path.exists(path):
        with open(path, "r", encoding="utf-8") as f:
            results = json.load(f)
    for name, fx, kind, kwargs, spec in gc.DECODE_CASES:
        if flt and flt not in name:
            continue
        root = fixture_root(fx)
        _model, decoder, decoder_dynamic = import_reference(root)
        with contextlib.redirect_stdout(io.StringIO()):
            dec = decoder_dynamic.DynamicDecoder(1) if kind == "dynamic" else decoder.Decoder(1)
        sents = gc.case_sentences(spec, _alpha[fx])
        store = _install_trace(dec, kind)
        t0 = time.time()
        case = []
        for si, s in enumerate(sents):
            if kwargs.get("random_sampling"):
                np.random.seed(gc.RANDOM_SAMPLING_SEED + si)
            res = dec.decode(s, **kwargs)
            item = {"input": s, "nbest": [[float(sc), list(ws)] for sc, ws in res]}
            if si < gc.TRACE_SENTENCES:
                fr = store["frame"]
                if kind == "dynamic":
                    item["trace"] = [store["snaps"].get(i, _snap(fr[i])) for i in range(len(s) + 1)]
                else:
                    item["trace"] = [_snap(fr[i]) for i in range(len(s) + 1)]
            store["snaps"] = {}
            case.append(item)
        results[name] = case
        print("decode", name, len(sents), "sentences %.1fs" % (time.time() - t0), "best", case[0]["nbest"][0][0])
    with open(path, "w", encoding="utf-8") as f:
        json.dump(results, f, ensure_ascii=False, indent=0)


def gen_eval():
    results = {}
    for name, fx, argv in gc.EVAL_CASES:
        root = fixture_root(fx)
        cfg, lexicon, _rd, _al = synth.build_fixture(root, fx)
        synth.write_test_corpus(root, lexicon, cfg["vocab_size"], **gc.EVAL_CORPUS)
        import_reference(root)
        work = tempfile.mkdtemp()
        os.makedirs(os.path.join(work, "eval"))
        cwd = os.getcwd()
        os.chdir(work)
        old_argv = sys.argv
        sys.argv = ["eval.py"] + argv
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                runpy.run_path(os.path.join(REF, "decoder", "eval.py"), run_name="__main__")
        finally:
            sys.argv = old_argv
            os.chdir(cwd)
        logs = os.listdir(os.path.join(work, "eval"))
        assert len(logs) == 1
        with open(os.path.join(work, "eval", logs[0]), "r", encoding="utf-8") as f:
            body = f.read()
        # timing text is not reproducible: keep everything before the summary's timings
        cut = body.index("--- ") if "--- " in body else len(body)
        hits = [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")]
        results[name] = {"log_name": logs[0], "log_body": body[:cut], "stdout_hits": hits}
        print("eval", name, hits)
    with open(os.path.join(GOLD, "eval.json"), "w", encoding="utf-8") as f:
        json.dump(results, f, ensure_ascii=False, indent=0)


def _run_reference_eval(root, argv):
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "eval"))
    cwd = os.getcwd()
    os.chdir(work)
    old_argv = sys.argv
    sys.argv = ["eval.py"] + argv
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            runpy.run_path(os.path.join(REF, "decoder", "eval.py"), run_name="__main__")
    finally:
        sys.argv = old_argv
        os.chdir(cwd)
    logs = os.listdir(os.path.join(work, "eval"))
    assert len(logs) == 1
    with open(os.path.join(work, "eval", logs[0]), "r", encoding="utf-8") as f:
        body = f.read()
    cut = body.index("--- ") if "--- " in body else len(body)
    hits = [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")]
    return {"log_name": logs[0], "log_body": body[:cut], "stdout_hits": hits}


def gen_ngram():
    """The n-gram baseline (reference decoder/decoder_ngram.py + model_ngram.py) on data/lm3 of the fixture."""
    results = {}
    for name, fx, order, kwargs, spec in gc.NGRAM_CASES:
        root = fixture_root(fx)
        import_reference(root)
        sys.modules.pop("decoder_ngram", None)
        sys.modules.pop("model_ngram", None)
        with contextlib.redirect_stdout(io.StringIO()):
            import decoder_ngram
            dec = decoder_ngram.NGramDecoder(1, ngram_order=order)
        sents = gc.ngram_sentences(spec, _alpha[fx], name == gc.NGRAM_CASES[0][0])
        out = []
        for s in sents:
            nbest = dec.decode(s, **kwargs)
            item = {"input": s, "nbest": [[float(a), list(b)] for a, b in nbest]}
            if nbest:
                item["evaluate_best"] = float(dec.model.evaluate(list(nbest[0][1])))
            out.append(item)
        results[name] = out
        print("ngram", name, len(out), "sentences;", sum(1 for o in out if not o["nbest"]), "without a path")
    name, fx, argv = gc.NGRAM_EVAL_CASE
    root = fixture_root(fx)
    cfg, lexicon, _rd, _al = synth.build_fixture(root, fx)
    synth.write_test_corpus(root, lexicon, cfg["vocab_size"], **gc.EVAL_CORPUS)
    import_reference(root)
    results[name] = _run_reference_eval(root, argv)
    print("ngram eval", results[name]["stdout_hits"])
    with open(os.path.join(GOLD, "ngram.json"), "w", encoding="utf-8") as f:
        json.dump(results, f, ensure_ascii=False, indent=0)


def gen_char():
    """The reference's CharRNNDecoder (decoder/decoder.py:244-341) dies on its first lattice look-up: ``_check_oov`` reads
    ``self.vocab.words`` (:263-264), which no Vocab defines, and ``Decoder._load_vocab`` (:70-73) gives it the word index where its
    character steps need ``CharVocab.c2i``.  Nothing of the reference is edited or copied: a subclass created HERE, at run time,
    supplies ``_load_vocab`` -- ``vocab`` a CharVocab, ``vocab.words`` its word index, ``w2i`` / ``i2w`` its character index -- and
    every other statement that runs is the reference's own."""
    results = {}
    for name, fx, kwargs, spec in gc.CHAR_CASES:
        root = fixture_root(fx)
        _model, decoder, _dd = import_reference(root)
        from train.data import CharVocab

        class Wired(decoder.CharRNNDecoder):
            def _load_vocab(self):
                self.vocab = CharVocab(self.config['vocab_size'])
                self.vocab.words = self.vocab.w2i
                self.w2i = self.vocab.c2i
                self.i2w = self.vocab.i2c

        with contextlib.redirect_stdout(io.StringIO()):
            dec = Wired(1)
        store = {}
        orig = dec._build_current_frame

        def wrapped(nodes, frame, idx, _orig=orig, _store=store):
            _store["frame"] = frame
            return _orig(nodes, frame, idx)

        dec._build_current_frame = wrapped
        sents = gc.case_sentences(spec, _alpha[fx])
        t0 = time.time()
        case = []
        for si, s in enumerate(sents):
            res = dec.decode(s, **kwargs)
            item = {"input": s, "nbest": [[float(sc), list(ws)] for sc, ws in res]}
            if si < gc.TRACE_SENTENCES:
                item["trace"] = [_snap(store["frame"][i]) for i in range(len(s) + 1)]
            case.append(item)
        results[name] = case
        print("char", name, len(sents), "sentences %.1fs" % (time.time() - t0), "best", case[0]["nbest"][0][0])
    # eval.py run unchanged; the class it imports (eval.py:7) carries the same ``_load_vocab``, set as an attribute at run time
    name, fx, argv = gc.CHAR_EVAL_CASE
    root = fixture_root(fx)
    cfg, lexicon, _rd, _al = synth.build_fixture(root, fx)
    synth.write_test_corpus(root, lexicon, cfg["vocab_size"], **gc.EVAL_CORPUS)
    _model, decoder, _dd = import_reference(root)
    from train.data import CharVocab

    def _load_vocab(self):
        self.vocab = CharVocab(self.config['vocab_size'])
        self.vocab.words = self.vocab.w2i
        self.w2i = self.vocab.c2i
        self.i2w = self.vocab.i2c

    # (a subclass under the module's name would recurse in the class's own super() call: the METHOD is set on the class instead)
    decoder.CharRNNDecoder._load_vocab = _load_vocab
    results[name] = _run_reference_eval(root, argv)
    print("char eval", results[name]["stdout_hits"])
    with open(os.path.join(GOLD, "char.json"), "w", encoding="utf-8") as f:
        json.dump(results, f, ensure_ascii=False, indent=0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--filter", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if a.only in (None, "lm"):
        gen_lm()
    if a.only in (None, "decode"):
        gen_decode(a.filter)
    if a.only in (None, "eval"):
        gen_eval()
    if a.only in (None, "ngram"):
        gen_ngram()
    if a.only in (None, "char"):
        gen_char()
