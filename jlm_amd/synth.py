"""Seeded synthetic artefacts in the reference's on-disk formats.

The reference ships no weights, lexicon or test data ("zero
data/weights"), and its corpus tooling needs the licensed BCCWJ corpus, so the
build generates stand-ins with the same file formats:

* ``data/lexicon.pkl``      list of ``(word, freq)`` sorted by (-freq, word)
                            (reference data.py:33,44); ``word`` is
                            ``display/reading/POS`` or ``<eos>``.
* ``data/reading_dict.pkl`` dict reading -> list of lexicon indices
                            (reference data.py:57-76,86).
* ``data/test.txt``         one sentence per line, tokens separated by a blank
                            (read by reference decoder/eval.py:141-166).
* ``train/experiments/<id>/config.json``   keys used at inference
                            (reference decoder/model.py:39-71).
* ``train/experiments/<id>/weights/lstm_weights.pkl`` dict of float32 arrays
                            (reference train/weights.py:30-74).

Everything is drawn from ``numpy.random.RandomState(seed)`` whose stream is
stable across numpy versions, so the same seeds regenerate the same artefacts
in this container and on the GPU box; only *outputs* are committed as golden
vectors.  Parameters follow SURVEY.md section 8(d).
"""
import json
import os
import pickle

import numpy as np

KANA = [chr(0x30A1 + i) for i in range(80)]
READING_LEN_P = (0.05, 0.30, 0.40, 0.25)
README_SEGS = [(200, 0, 12000), (100, 12000, 30000), (50, 30000, None)]


def make_lexicon(vocab_size, seed=1234, oov_frac=0.1, alphabet=80, display_alphabet=0):
    """-> (lexicon, reading_dict).  ``<eos>`` is the most frequent entry; the
    first ``vocab_size-1`` entries are in-vocabulary, the ``oov_frac`` tail is
    not (exercises the skip at reference decoder/decoder.py:99-103).
    ``display_alphabet`` > 0 (the character-model fixtures): the display string of a word is 1-3 characters of that many CJK code
    points (frequent ones more often), so that a character vocabulary (reference train/data.py:28-47) is worth a softmax and several
    words of a reading share a display string (the ``word_set`` dedup at decoder.py:116-122); 0: ``w<i>`` as before."""
    rng = np.random.RandomState(seed)
    n_words = int(round(vocab_size * (1.0 + oov_frac)))
    lens = rng.choice([1, 2, 3, 4], size=n_words, p=READING_LEN_P)
    chars = rng.randint(0, alphabet, size=(n_words, 4))
    top = n_words + 10
    lexicon = [("<eos>", top + 1)]
    if display_alphabet:
        rng2 = np.random.RandomState(seed + 4321)          # a stream of its own: the readings stay those of the word fixtures
        dlens = rng2.choice([1, 2, 3], size=n_words, p=(0.3, 0.45, 0.25))
        dchars = np.minimum((rng2.rand(n_words, 3) ** 4 * display_alphabet).astype(int), display_alphabet - 1)
    for i in range(n_words):
        reading = "".join(KANA[c] for c in chars[i, : lens[i]])
        disp = "".join(chr(0x4E00 + c) for c in dchars[i, : dlens[i]]) if display_alphabet else "w%d" % i
        lexicon.append(("%s/%s/N" % (disp, reading), top - i))
    reading_dict = {}
    for i, (word, _) in enumerate(lexicon):
        tokens = word.split("/")
        if len(tokens) < 3:
            continue
        reading = tokens[1] if tokens[1] != "" else tokens[0]
        reading_dict.setdefault(reading, []).append(i)
    return lexicon, reading_dict


def char_index(lexicon, vocab_size):
    """character -> index over the in-vocabulary words' display strings, as the reference's CharVocab builds it (train/data.py:28-40):
    ``<unk>`` 0, ``<eos>`` 1, then first occurrence order over Vocab.lexicon[2:]"""
    c2i = {"<unk>": 0, "<eos>": 1}
    for word, _f in ([("<unk>", 0)] + list(lexicon[: vocab_size - 1]))[2:]:
        for c in word.split("/")[0]:
            if c not in c2i:
                c2i[c] = len(c2i)
    return c2i


def write_lexicon(root, vocab_size, seed=1234, oov_frac=0.1, alphabet=80, display_alphabet=0):
    lexicon, reading_dict = make_lexicon(vocab_size, seed, oov_frac, alphabet, display_alphabet)
    d = os.path.join(root, "data")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "lexicon.pkl"), "wb") as f:
        pickle.dump(lexicon, f)
    with open(os.path.join(d, "reading_dict.pkl"), "wb") as f:
        pickle.dump(reading_dict, f)
    return lexicon, reading_dict


def make_config(vocab_size, hidden, embed, mode="tied", segs=None, self_norm=False, char_rnn=False):
    """config.json content.  ``mode``: tied | untied | dsoftmax | vtable.  ``char_rnn``: the model's softmax runs over characters
    (``vocab_size`` stays the WORD vocabulary's size: reference train/model.py:95-96, decoder/eval.py:35-36)."""
    assert mode in ("tied", "untied", "dsoftmax", "vtable")
    cfg = {
        "vocab_size": int(vocab_size),
        "hidden_size": int(hidden),
        "embed_size": int(embed),
        "share_embedding": mode != "untied",
        "D_softmax": mode == "dsoftmax",
        "V_table": mode == "vtable",
        "embedding_seg": [list(s) for s in (segs if segs is not None else README_SEGS)],
        "self_norm": bool(self_norm),
        "char_rnn": bool(char_rnn),
    }
    return cfg


def make_weights(cfg, seed=7, scale=0.05, n_out=None):
    """Weight dict with the key names / shapes of reference train/weights.py:30-55
    (shapes from reference train/model.py:137-150,188-193,57-60).  ``n_out``: rows of the softmax when it is not the word
    vocabulary (character models: len(CharVocab))."""
    rng = np.random.RandomState(seed)

    def w(*shape):
        return rng.normal(0.0, scale, size=shape).astype(np.float32)

    V, H = (cfg["vocab_size"] if n_out is None else int(n_out)), cfg["hidden_size"]
    segs = cfg["embedding_seg"]
    if cfg["V_table"]:
        E = segs[0][0]                       # reference train/model.py:53
    elif cfg["D_softmax"]:
        E = sum(s[0] for s in segs)          # reference train/model.py:77
    else:
        E = cfg["embed_size"]
    weights = {}
    for g in "ifog":
        weights["HM" + g] = w(H, H)
    for g in "ifog":
        weights["IM" + g] = w(E, H)
    for g in "ifog":
        weights["b" + g] = w(H)
    weights["b2"] = w(V)
    if cfg["share_embedding"]:
        weights["PM"] = w(H, E)
    else:
        weights["UM"] = w(H, V)
    if cfg["V_table"]:
        for i, (size, s, e) in enumerate(segs):
            e = V if e is None else e
            weights["LM%d" % i] = w(e - s, size)
            if i != 0:
                weights["VT%d" % i] = w(size, E)
    elif cfg["D_softmax"]:
        blocks = []
        for size, s, e in segs:
            e = V if e is None else e
            blocks.append(w(e - s, size))
        weights["LM"] = blocks
    else:
        weights["LM"] = w(V, E)
    return weights


def _output_blocks(weights):
    """(key, index-or-None) of every output-embedding block of a weight dict (LM, the D_softmax block list, LM0..)"""
    out = []
    for key in sorted(weights):
        if key.startswith("LM"):
            if isinstance(weights[key], list):
                out += [(key, i) for i in range(len(weights[key]))]
            else:
                out.append((key, None))
    return out


def shape_weights(weights, cfg, shape, seed=7):
    """Statistics a TRAINED model has and N(0, 0.05^2) draws do not (the reference ships no weights; its claims are about trained
    language models, README.md:6,15,72).  In place, on top of make_weights' draws, from a stream of its own:

      peaked<N>  output embeddings x N (N = 10 if absent): logits of +-N (peaked next-word distributions), and a unigram-like bias
                 b2[w] = -log(w + 8) (the lexicon is sorted by frequency, reference data.py:33,44: frequent words first)
      heavy      output-embedding blocks redrawn from Student-t(3) at the same standard deviation: max|B| / rms B in the hundreds
                 (outlier rows and columns, as trained embeddings have), same unigram-like bias
    """
    if not shape:
        return weights
    rng = np.random.RandomState(seed + 1000)
    V = cfg["vocab_size"]
    if shape.startswith("peaked"):
        mult = np.float32(float(shape[6:]) if len(shape) > 6 else 10.0)
        for key, i in _output_blocks(weights):
            if i is None:
                weights[key] = weights[key] * mult
            else:
                weights[key][i] = weights[key][i] * mult
    elif shape == "heavy":
        for key, i in _output_blocks(weights):
            blk = weights[key] if i is None else weights[key][i]
            std = float(blk.std())
            new = (rng.standard_t(3, size=blk.shape) * (std / np.sqrt(3.0))).astype(np.float32)
            if i is None:
                weights[key] = new
            else:
                weights[key][i] = new
    else:
        raise ValueError(shape)
    weights["b2"] = (-np.log(np.arange(V, dtype=np.float64) + 8.0)).astype(np.float32)
    return weights


def write_experiment(root, exp_id, cfg, seed=7, scale=0.05, shape=None, n_out=None):
    d = os.path.join(root, "train", "experiments", str(exp_id))
    os.makedirs(os.path.join(d, "weights"), exist_ok=True)
    with open(os.path.join(d, "config.json"), "wt") as f:
        f.write(json.dumps(cfg))
    weights = shape_weights(make_weights(cfg, seed, scale, n_out), cfg, shape, seed)
    with open(os.path.join(d, "weights", "lstm_weights.pkl"), "wb") as f:
        pickle.dump(weights, f)
    return weights


def make_sentences(n, length, seed=99, alphabet=80):
    """Uniform random kana strings of the stated length."""
    rng = np.random.RandomState(seed)
    idx = rng.randint(0, alphabet, size=(n, length))
    return ["".join(KANA[c] for c in row) for row in idx]


def make_ragged_sentences(n, min_len, max_len, seed=99, alphabet=80):
    rng = np.random.RandomState(seed)
    lens = rng.randint(min_len, max_len + 1, size=n)
    return ["".join(KANA[c] for c in rng.randint(0, alphabet, size=l)) for l in lens]


def make_test_corpus(lexicon, vocab_size, n, words_per_sentence=6, seed=5, oov_every=0):
    """Lines for data/test.txt: concatenations of random in-vocabulary words so
    that an eval target exists.  ``oov_every``>0 inserts an
    out-of-vocabulary word into every such line (those lines are skipped by
    the harness, reference decoder/eval.py:131-143)."""
    rng = np.random.RandomState(seed)
    lines = []
    for j in range(n):
        ids = rng.randint(1, vocab_size - 1, size=words_per_sentence)
        toks = [lexicon[i][0] for i in ids]
        if oov_every and j % oov_every == oov_every - 1:
            toks[len(toks) // 2] = lexicon[len(lexicon) - 1 - (j % 7)][0]
        lines.append(" ".join(toks))
    return lines


def write_test_corpus(root, lexicon, vocab_size, n, words_per_sentence=6, seed=5, oov_every=0):
    lines = make_test_corpus(lexicon, vocab_size, n, words_per_sentence, seed, oov_every)
    d = os.path.join(root, "data")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "test.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    return lines


# Named fixture recipes shared by tests, golden-vector generation and bench.
def small_segs(V):
    return [(32, 0, V // 4), (16, V // 4, (3 * V) // 5), (8, (3 * V) // 5, None)]


def wide_segs(V):
    """BASELINE configs[1]'s segment widths (200 / 100 / 50) over a small vocabulary: the shapes the normaliser runs on mixed rows"""
    return [(200, 0, V // 4), (100, V // 4, (3 * V) // 5), (50, (3 * V) // 5, None)]


def wideh_segs(V):
    """200 / 100 / 36: two mixed-row shapes and a short segment the hybrid launch keeps on split rows"""
    return [(200, 0, V // 4), (100, V // 4, (3 * V) // 5), (36, (3 * V) // 5, None)]


def write_arpa(root, lexicon, vocab_size, seed=77, name="lm3"):
    """A synthetic back-off n-gram file ``data/lm3`` in the layout the reference's parser reads
    (decoder/model_ngram.py:30-52: tab-separated ``log10 prob <TAB> w1 w2 .. [<TAB> log10 backoff]``, every other line
    ignored; ``<s>`` / ``</s>`` stand for ``<eos>``).  Unigrams for ~80 % of the in-vocabulary words (the rest fall
    to the 100.0 floor), bigrams and trigrams over random word pairs / triples, some anchored at the sentence start."""
    rng = np.random.RandomState(seed)
    words = [w for w, _f in lexicon[1:vocab_size - 1]]
    n = len(words)
    fmt = lambda x: "%.4f" % x
    lines = ["", "\\data\\", "ngram 1=0", "ngram 2=0", "ngram 3=0", "", "\\1-grams:"]
    lines.append("%s\t</s>" % fmt(-1.2))
    lines.append("%s\t<s>\t%s" % (fmt(-99.0), fmt(-0.3)))
    keep = rng.rand(n) < 0.8
    lp = -1.0 - 4.0 * rng.rand(n)
    bo = -0.1 - 0.8 * rng.rand(n)
    for i in range(n):
        if keep[i]:
            lines.append("%s\t%s\t%s" % (fmt(lp[i]), words[i], fmt(bo[i])))
    lines += ["", "\\2-grams:"]
    for _ in range(4 * n):
        a, b = rng.randint(0, n, size=2)
        first = "<s>" if rng.rand() < 0.1 else words[a]
        last = "</s>" if rng.rand() < 0.05 else words[b]
        lines.append("%s\t%s %s\t%s" % (fmt(-0.3 - 3.0 * rng.rand()), first, last, fmt(-0.1 - 0.5 * rng.rand())))
    lines += ["", "\\3-grams:"]
    for _ in range(3 * n):
        a, b, c = rng.randint(0, n, size=3)
        first = "<s>" if rng.rand() < 0.1 else words[a]
        lines.append("%s\t%s %s %s" % (fmt(-0.2 - 2.5 * rng.rand()), first, words[b], words[c]))
    lines += ["", "\\end\\", ""]
    d = os.path.join(root, "data")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w", encoding="utf-8") as f:
        f.write("\n".join(lines))


def build_fixture(root, name, exp_id=1):
    """Materialise one named fixture under ``root``; returns (cfg, lexicon,
    reading_dict, alphabet).  The small fixtures use a 12-kana alphabet so
    that their 2 200-word lexicon still gives a dense lattice.  Names:

      small-{tied,untied,dsoftmax,vtable}[-sn]   V=2000 H=64 E=32 (unit tests)
      small-char / mid-char                       character models (config char_rnn): tied softmax over the characters of the
                                                  in-vocabulary words' display strings (300 / 3 000 CJK code points), same lexicon sizes
      wide-{vtable,dsoftmax}                      V=2000 H=64, segments 200 / 100 / 50 (the mixed-row shapes, small vocabulary)
      wideh-vtable                                 the same with 200 / 100 / 36: the hybrid launch (mixed + split rows)
      wide128-tied                                 tied, E = 128: mixed rows without bias columns (the form of the tied k = 256 models)
      mid-tied / mid-vtable / mid-untied          V=50000 H=512 (configs 1 / 2; untied projection UM [H, V])
      big-tied                                    V=100000 H=512 E=256 (config 3)
      bigpeaked-tied                              config 3's model with output embeddings x 10 and the unigram-like bias
      peaked-{vtable,tied} / peaked20-{vtable,tied} / heavy-{vtable,tied}
                                                  the mid-* models with trained-model-like output embeddings (shape_weights): logits
                                                  of +-10 / +-20 and a unigram-like bias; Student-t(3) blocks
    """
    parts = name.split("-")
    size, mode = parts[0], parts[1]
    self_norm = len(parts) > 2 and parts[2] == "sn"
    alphabet, scale, shape = 80, 0.05, None
    display_alphabet = 0
    if mode == "char":                    # a character model (tied softmax over CharVocab) decoded on the word lattice
        mode, display_alphabet = "tied", {"small": 300, "mid": 3000}[size]
    if size in ("peaked", "peaked20", "heavy"):
        shape, size = size, "mid"
    elif size == "bigpeaked":             # config 3's model (V = 100 k) with the peaked statistics
        shape, size = "peaked", "big"
    if size == "small":
        scale = 0.25                      # keeps the tiny model's logits O(1)
        V, H, E, segs, alphabet = 2000, 64, 32, small_segs(2000), 12
    elif size == "wide":
        scale = 0.1
        V, H, E, segs, alphabet = 2000, 64, 200, wide_segs(2000), 12
    elif size == "wide128":             # tied, embedding 128: a contraction that fills its last block (biases outside the rows)
        scale = 0.12
        V, H, E, segs, alphabet = 2000, 64, 128, small_segs(2000), 12
    elif size == "wideh":
        scale = 0.1
        V, H, E, segs, alphabet = 2000, 64, 200, wideh_segs(2000), 12
    elif size == "mid":
        V, H, E, segs = 50000, 512, 256, README_SEGS
    elif size == "big":
        V, H, E, segs = 100000, 512, 256, README_SEGS
    else:
        raise ValueError(name)
    cfg = make_config(V, H, E, mode, segs, self_norm, char_rnn=bool(display_alphabet))
    lexicon, reading_dict = write_lexicon(root, V, alphabet=alphabet, display_alphabet=display_alphabet)
    n_out = len(char_index(lexicon, V)) if display_alphabet else None
    write_experiment(root, exp_id, cfg, scale=scale, shape=shape, n_out=n_out)
    if size == "small":
        write_arpa(root, lexicon, V)      # the n-gram baseline's model file (decoder/model_ngram.py reads data/lm3)
    return cfg, lexicon, reading_dict, alphabet


def write_compressed(root, exp_id, bit=8, seed=3, formats=("pkl", "dump", "txt")):
    """k-means-style compressed copies of an experiment's weights in the formats of
    reference train/comp.py:52-80.  The codebook here is a quantile grid (sklearn's KMeans
    is slow and its n_jobs argument no longer exists); the FILE FORMATS are the point:
    code uint8 with the tensor's shape, codebook float32 [2**bit, 1].
    -> dict name -> decoded array (what inference must see)."""
    d = os.path.join(root, "train", "experiments", str(exp_id), "weights")
    with open(os.path.join(d, "lstm_weights.pkl"), "rb") as f:
        weights = pickle.load(f)
    cdir = os.path.join(d, "comp_{}".format(bit))
    os.makedirs(cdir, exist_ok=True)
    decoded, dump = {}, {}
    for k, v in weights.items():
        if isinstance(v, list):
            raise ValueError("train/comp.py cannot compress the D_softmax block list either")
        flat = v.reshape(-1).astype(np.float64)
        book = np.quantile(flat, (np.arange(2 ** bit) + 0.5) / 2 ** bit).astype(np.float32).reshape(-1, 1)
        mids = (book[1:, 0].astype(np.float64) + book[:-1, 0]) / 2
        code = np.searchsorted(mids, flat).astype(np.uint8 if bit <= 8 else np.int64).reshape(v.shape)
        dump[k] = (code, book)
        decoded[k] = np.take(book, code)
        if "txt" in formats:
            np.savetxt(os.path.join(cdir, "{}_code.txt".format(k)), code.astype(int), fmt="%i")
            np.savetxt(os.path.join(cdir, "{}_codebook.txt".format(k)), book)
    if "pkl" in formats:
        with open(os.path.join(d, "lstm_weights_comp_{}.pkl".format(bit)), "wb") as f:
            pickle.dump(decoded, f)
    if "dump" in formats:
        with open(os.path.join(cdir, "lstm_weights_comp_dump.pkl"), "wb") as f:
            pickle.dump(dump, f)
    return decoded


def write_verbose_dumps(root, exp_id, npy=True):
    """The .txt/.npy per-tensor dumps of reference train/weights.py:76-87."""
    d = os.path.join(root, "train", "experiments", str(exp_id), "weights")
    with open(os.path.join(d, "lstm_weights.pkl"), "rb") as f:
        weights = pickle.load(f)
    for name, m in weights.items():
        if isinstance(m, list):
            for i, item in enumerate(m):
                np.savetxt(os.path.join(d, "{}{}.txt".format(name, i)), item)
        else:
            np.savetxt(os.path.join(d, name + ".txt"), m)
            if npy:
                np.save(os.path.join(d, name + ".npy"), m)
