"""Case list shared by tools/make_golden.py (which runs the REFERENCE in the
build container) and the tests (which replay the same seeded inputs through
the oracle / the HIP path).  Only outputs are committed under tests/golden/."""
import numpy as np

from jlm_amd import synth

LM_FIXTURES = ["small-tied", "small-untied", "small-dsoftmax", "small-vtable", "small-tied-sn",
               "mid-tied", "mid-vtable",
               # trained-model-like output embeddings (synth.shape_weights): logits of +-10 / +-20 and a unigram-like bias; Student-t(3) blocks
               "peaked-vtable", "peaked20-vtable", "peaked-tied", "peaked20-tied", "heavy-vtable"]
SHAPED_FIXTURES = [f for f in LM_FIXTURES if f.split("-")[0] in ("peaked", "peaked20", "heavy")]
LM_ROWS = [1, 10]
LM_STEPS = 3
LM_NCOLS = 256
LM_SUBSET = 300


def lm_inputs(cfg, rows, seed=11):
    """index lists for LM_STEPS consecutive steps, the sorted vocab subset and
    the sampled logit columns."""
    rng = np.random.RandomState(seed + rows)
    V = cfg["vocab_size"]
    idx = [[int(x) for x in rng.randint(0, V, size=rows)] for _ in range(LM_STEPS)]
    subset = sorted(int(x) for x in rng.choice(V, size=LM_SUBSET, replace=False))
    cols = np.sort(rng.choice(V, size=LM_NCOLS, replace=False))
    h0 = rng.normal(0, 0.3, size=(rows, cfg["hidden_size"]))
    c0 = rng.normal(0, 0.3, size=(rows, cfg["hidden_size"]))
    return idx, subset, cols, h0, c0


# (case name, fixture, decoder kind, decode kwargs, sentence spec)
# sentence spec: ("ragged", n, min_len, max_len, seed) | ("fixed", n, length, seed)
DECODE_CASES = [
    ("small-tied/static", "small-tied", "static", dict(beam_width=10), ("ragged", 12, 1, 20, 3)),
    ("small-tied/static-b3", "small-tied", "static", dict(beam_width=3, topN=5), ("ragged", 8, 1, 16, 4)),
    ("small-tied/static-vs", "small-tied", "static", dict(beam_width=10, vocab_select=True), ("ragged", 12, 1, 20, 3)),
    ("small-tied/static-vs-top", "small-tied", "static",
     dict(beam_width=10, vocab_select=True, samples=50, top_sampling=True), ("ragged", 6, 2, 14, 5)),
    ("small-tied/static-vs-rand", "small-tied", "static",
     dict(beam_width=10, vocab_select=True, samples=30, random_sampling=True), ("ragged", 6, 2, 14, 6)),
    ("small-tied-sn/static", "small-tied-sn", "static", dict(beam_width=10), ("ragged", 8, 1, 16, 7)),
    ("small-untied/static", "small-untied", "static", dict(beam_width=10), ("ragged", 8, 1, 16, 8)),
    ("small-dsoftmax/static", "small-dsoftmax", "static", dict(beam_width=10), ("ragged", 8, 1, 16, 9)),
    ("small-dsoftmax/static-vs", "small-dsoftmax", "static", dict(beam_width=10, vocab_select=True), ("ragged", 8, 1, 16, 9)),
    ("small-vtable/static", "small-vtable", "static", dict(beam_width=10), ("ragged", 8, 1, 16, 10)),
    ("small-vtable/static-vs", "small-vtable", "static", dict(beam_width=10, vocab_select=True), ("ragged", 8, 1, 16, 10)),
    ("small-tied/dynamic", "small-tied", "dynamic", dict(beam_width=10, vocab_select=True), ("ragged", 12, 1, 20, 3)),
    ("small-tied/dynamic-b4", "small-tied", "dynamic", dict(beam_width=4, vocab_select=True), ("ragged", 8, 1, 16, 4)),
    ("small-tied/dynamic-top", "small-tied", "dynamic",
     dict(beam_width=10, vocab_select=True, samples=20, top_sampling=True), ("ragged", 6, 2, 14, 5)),
    ("small-tied-sn/dynamic", "small-tied-sn", "dynamic", dict(beam_width=10, vocab_select=True), ("ragged", 8, 1, 16, 7)),
    # round 6: random ids appended to the frame-0 list (decoder_dynamic.py:38-41: NOT de-duplicated there), np.random seeded per sentence as
    # for static-vs-rand; and the unpruned incremental search (decoder_dynamic.py:89-91: no sort, no cut) on inputs short enough for it
    ("small-tied/dynamic-rand", "small-tied", "dynamic",
     dict(beam_width=10, vocab_select=True, samples=30, random_sampling=True), ("ragged", 6, 2, 14, 6)),
    ("small-tied/dynamic-unpruned", "small-tied", "dynamic", dict(beam_width=None, vocab_select=True, topN=50), ("ragged", 6, 1, 3, 11)),
    # the incremental decoder on SEGMENTED models: the reference pairs weight rows and biases of different words there
    # (SURVEY 8 a16); jlm_amd reproduces it with DynamicDecoder.compat_quirks = True (the "-quirk" cases set it)
    ("small-vtable/dynamic-quirk", "small-vtable", "dynamic", dict(beam_width=10, vocab_select=True), ("ragged", 8, 1, 16, 10)),
    ("small-dsoftmax/dynamic-quirk", "small-dsoftmax", "dynamic", dict(beam_width=10, vocab_select=True), ("ragged", 8, 1, 16, 9)),
    ("small-vtable/dynamic-quirk-top", "small-vtable", "dynamic",
     dict(beam_width=5, vocab_select=True, samples=20, top_sampling=True), ("ragged", 6, 2, 14, 5)),
    # BASELINE.json configs[0]: 100 sentences, beam 10, V=50k, tied softmax
    ("mid-tied/static", "mid-tied", "static", dict(beam_width=10), ("fixed", 100, 20, 99)),
    ("mid-tied/static-vs", "mid-tied", "static", dict(beam_width=10, vocab_select=True), ("fixed", 24, 20, 98)),
    ("mid-tied/dynamic", "mid-tied", "dynamic", dict(beam_width=10, vocab_select=True), ("fixed", 24, 20, 98)),
    ("mid-vtable/static", "mid-vtable", "static", dict(beam_width=10), ("fixed", 24, 20, 97)),
    ("mid-vtable/dynamic-quirk", "mid-vtable", "dynamic", dict(beam_width=10, vocab_select=True), ("fixed", 8, 20, 94)),
    ("big-tied/static-b20", "big-tied", "static", dict(beam_width=20), ("fixed", 16, 20, 96)),
    # untied projection (model.py:189-191) at BASELINE size: V=50k, k = H = 512 (the tile-form normaliser inside the frame loop)
    ("mid-untied/static", "mid-untied", "static", dict(beam_width=10), ("fixed", 24, 20, 95)),
    # BASELINE-size models with trained-model-like statistics (round 4): what the int8 cross terms of the mixed-row normaliser
    # are sensitive to.  The default path must hold the same bars on them (DeviceModel._calibrate_mixed decides the form).
    ("peaked-vtable/static", "peaked-vtable", "static", dict(beam_width=10), ("fixed", 24, 20, 93)),
    ("peaked20-vtable/static", "peaked20-vtable", "static", dict(beam_width=10), ("fixed", 24, 20, 92)),
    ("peaked-tied/static", "peaked-tied", "static", dict(beam_width=10), ("fixed", 24, 20, 91)),
    ("peaked20-tied/static", "peaked20-tied", "static", dict(beam_width=10), ("fixed", 24, 20, 90)),
    ("heavy-vtable/static", "heavy-vtable", "static", dict(beam_width=10), ("fixed", 24, 20, 89)),
    ("peaked20-vtable/static-vs", "peaked20-vtable", "static", dict(beam_width=10, vocab_select=True), ("fixed", 8, 20, 88)),
    ("peaked20-tied/dynamic", "peaked20-tied", "dynamic", dict(beam_width=10, vocab_select=True), ("fixed", 8, 20, 87)),
    # round 5: the stated sentence lengths L = 10 and L = 40 -- a path score is a sum over the frames, so its error grows
    # with L -- and trained-like weights for the selected-vocabulary decoders and for configs[2]'s shape
    ("mid-vtable/static-L40", "mid-vtable", "static", dict(beam_width=10), ("fixed", 16, 40, 86)),
    ("peaked-vtable/static-L40", "peaked-vtable", "static", dict(beam_width=10), ("fixed", 16, 40, 85)),
    ("peaked20-tied/static-L40", "peaked20-tied", "static", dict(beam_width=10), ("fixed", 8, 40, 84)),
    ("mid-tied/dynamic-L40", "mid-tied", "dynamic", dict(beam_width=10, vocab_select=True), ("fixed", 8, 40, 83)),
    ("mid-vtable/static-L10", "mid-vtable", "static", dict(beam_width=10), ("fixed", 16, 10, 82)),
    ("mid-tied/dynamic-L10", "mid-tied", "dynamic", dict(beam_width=10, vocab_select=True), ("fixed", 8, 10, 81)),
    ("peaked-tied/dynamic", "peaked-tied", "dynamic", dict(beam_width=10, vocab_select=True), ("fixed", 16, 20, 80)),
    ("peaked-vtable/static-vs", "peaked-vtable", "static", dict(beam_width=10, vocab_select=True), ("fixed", 16, 20, 79)),
    ("bigpeaked-tied/static-b20", "bigpeaked-tied", "static", dict(beam_width=20), ("fixed", 8, 20, 78)),
]

# Character models on the word lattice (reference CharRNNDecoder, decoder.py:244-341, with the one run-time wiring of
# tools/make_golden.py: parity unpinned): (case name, fixture, decode kwargs, sentence spec)
CHAR_CASES = [
    ("small-char/char", "small-char", dict(beam_width=10), ("ragged", 12, 1, 20, 3)),
    ("small-char/char-b3", "small-char", dict(beam_width=3, topN=5), ("ragged", 8, 1, 16, 4)),
    ("small-char/char-vs-top", "small-char", dict(beam_width=10, vocab_select=True, samples=20, top_sampling=True), ("ragged", 4, 2, 12, 5)),
    ("small-char/char-unpruned", "small-char", dict(beam_width=None, topN=50), ("ragged", 6, 1, 3, 11)),
    ("mid-char/char", "mid-char", dict(beam_width=10), ("fixed", 16, 20, 77)),
]


# ... and the reference's eval.py, unchanged, on the character fixture (decoder.CharRNNDecoder._load_vocab set at run time,
# tools/make_golden.py): decoder selection by config['char_rnn'] (eval.py:43-44), CharVocab for the eval set (eval.py:35-36)
CHAR_EVAL_CASE = ("small-char/eval-char", "small-char", ["-e", "1", "-es", "12", "-b", "10"])


def is_quirk_case(name):
    """cases that need ``compat_quirks = True`` on the jlm_amd decoder"""
    return name.split("/")[1].startswith("dynamic-quirk")


RANDOM_SAMPLING_SEED = 123
TRACE_SENTENCES = 4          # per case: frame-by-frame beams kept for the first few sentences


def case_sentences(spec, alphabet):
    if spec[0] == "ragged":
        _, n, lo, hi, seed = spec
        return synth.make_ragged_sentences(n, lo, hi, seed=seed, alphabet=alphabet)
    _, n, length, seed = spec
    return synth.make_sentences(n, length, seed=seed, alphabet=alphabet)


EVAL_CASES = [
    ("small-tied/eval-static", "small-tied", ["-e", "1", "-es", "20", "-b", "10"]),
    ("small-tied/eval-dynamic", "small-tied", ["-e", "1", "-es", "12", "-b", "10", "-vs", "True", "-dd", "True"]),
]
# n-gram baseline decoder (reference decoder/decoder_ngram.py, model data/lm3 written by synth.write_arpa):
# (name, fixture, ngram order, decode kwargs, sentence spec)
NGRAM_CASES = [
    ("small-tied/ngram-o3", "small-tied", 3, dict(beam_width=10), ("ragged", 12, 1, 20, 3)),
    ("small-tied/ngram-o2-b3", "small-tied", 2, dict(beam_width=3, topN=5), ("ragged", 8, 1, 16, 4)),
    ("small-tied/ngram-unpruned", "small-tied", 3, dict(beam_width=None, topN=10), ("ragged", 6, 1, 5, 5)),
]
NGRAM_UNCOVERED = "\u30f7"          # a kana no lexicon reading contains: this decoder has no <unk> fallback -> no path


def ngram_sentences(spec, alphabet, first_case):
    sents = case_sentences(spec, alphabet)
    if first_case:
        sents = sents + [sents[0][:3] + NGRAM_UNCOVERED + sents[0][3:]]
    return sents


NGRAM_EVAL_CASE = ("small-tied/eval-ngram", "small-tied", ["-e", "1", "-es", "20", "-b", "10", "-ng", "True", "-o", "3"])
EVAL_CORPUS = dict(n=30, words_per_sentence=5, seed=5, oov_every=6)
