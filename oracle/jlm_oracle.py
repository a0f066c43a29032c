"""CPU oracle: a numpy restatement of the reference's LSTM LM + lattice decoders.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``jlm_amd``) never does and fails loudly when the HIP
library is missing.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so this restatement is pinned against OUTPUTS OF THE REFERENCE
ITSELF, captured by importing /root/reference unmodified in the build
container (``tools/make_golden.py`` -> ``tests/golden/*.npz|json``) and
checked by ``tests/test_oracle_golden.py``.

The arithmetic of the path lives in numpy/OpenBLAS (third party, version not
pinned by the reference; here numpy 2.2.6 / OpenBLAS 0.3.29): ``np.dot``,
``np.exp``, ``np.amax``, ``np.sum``, ``np.tanh`` and ``math.log``.  The same
calls are used here with the same operand dtypes, so float32 weights meet
float64 state exactly where they do in the reference.

Structure (array based, not the reference's Path/Node objects):

  OracleLM             <- decoder/model.py LSTM_Model            (model.py:36-198)
  build_lattice        <- decoder/decoder.py Decoder._build_lattice (decoder.py:79-135)
  static_vocab         <- Decoder._build_lattice_vocab           (decoder.py:137-151)
  static_decode        <- Decoder.decode + _build_current_frame + _batch_predict
                                                                   (decoder.py:164-241)
  dynamic_vocab        <- DynamicDecoder._build_lattice_vocab    (decoder_dynamic.py:30-46)
  dynamic_decode       <- DynamicDecoder.decode/_build_current_frame/
                          _incremental_decode/_fix_neg_log        (decoder_dynamic.py:49-194)
  build_char_lattice   <- Decoder._build_lattice, config['char_rnn'] branch (decoder.py:105-124)
  char_decode          <- CharRNNDecoder.decode/_build_current_frame/_eval_frame (decoder.py:273-341)
  OracleDecoder / OracleDynamicDecoder / OracleCharRNNDecoder : file-loading wrappers with the
                          reference's class signatures (decoder.py:54-77).

PARITY UNPINNED for the character decoder: the reference's CharRNNDecoder cannot run as
shipped (its _check_oov reads ``self.vocab.words``, decoder.py:263-264, which no Vocab defines, and
Decoder._load_vocab gives it the WORD index where its character steps need CharVocab.c2i).  Its
golden vectors come from the reference class with ONE method supplied at run time by a subclass
inside tools/make_golden.py (``_load_vocab``: CharVocab, ``vocab.words`` = the word index,
``w2i`` = ``c2i``) -- the evidently intended wiring, every other statement the reference's own.
"""
import json
import math
import os
import pickle
import sys
import time

import numpy as np


# --------------------------------------------------------------------------- LM
def sigmoid(x):
    """reference model.py:12-13 (no clamp; exp overflow -> 0 is benign)."""
    return 1 / (np.exp(-x) + 1)


def softmax(w):
    """reference model.py:15-20: 1-D input is promoted to [1, n]."""
    assert w.ndim == 2 or w.ndim == 1
    w = w[None, :] if w.ndim == 1 else w
    e = np.exp(w - np.amax(w, axis=1, keepdims=True))
    return e / np.sum(e, axis=1, keepdims=True)


class OracleLM:
    """One LSTM step + vocabulary projection (reference model.py:36-198)."""

    GATES = "ifog"

    def __init__(self, config, weights):
        self.config = config
        w = dict(weights)
        self.hidden_size = config["hidden_size"]
        self.embed_size = config["embed_size"]
        self.share_embedding = config["share_embedding"]
        self.segs = [tuple(s) for s in config["embedding_seg"]]
        self.blocks = None
        self.v_tables = None
        if config["D_softmax"]:
            # model.py:47-54: block-diagonal LM, float64 because of np.zeros
            self.blocks = w["LM"]
            self.embed_size = sum(s[0] for s in self.segs)
            full = np.zeros((w["b2"].shape[0], self.embed_size))
            c0 = 0
            for i, (size, s, e) in enumerate(self.segs):
                full[s:e, c0:c0 + size] = self.blocks[i]
                c0 += size
            w["LM"] = full
        if config["V_table"]:
            # model.py:56-71: input embedding = concat(B0, B1.VT1, B2.VT2)
            self.blocks, self.v_tables, emb = [], [], []
            for i in range(len(self.segs)):
                blk = w["LM%d" % i]
                self.blocks.append(blk)
                if i != 0:
                    vt = w["VT%d" % i]
                    self.v_tables.append(vt)
                    emb.append(np.dot(blk, vt))
                else:
                    self.v_tables.append(None)
                    emb.append(blk)
            w["LM"] = np.concatenate(emb, axis=0)
        self.weights = w

    def zero_state(self, rows=1):
        return np.zeros((rows, self.hidden_size)), np.zeros((rows, self.hidden_size))

    def lstm_cell(self, index, hidden, cell):
        """reference model.py:125-139.  Gate order i, f, o, g."""
        w = self.weights
        e = w["LM"][index, :]
        z = {}
        for g in self.GATES:
            z[g] = np.dot(hidden, w["HM" + g]) + np.dot(e, w["IM" + g]) + w["b" + g]
        i, f, o = sigmoid(z["i"]), sigmoid(z["f"]), sigmoid(z["o"])
        g = np.tanh(z["g"])
        cell = np.multiply(cell, f) + np.multiply(g, i)
        hidden = np.multiply(np.tanh(cell), o)
        return hidden, cell

    def _seg_columns(self, vocab, s, e):
        if e is None:
            e = sys.maxsize
        return [v - s for v in vocab if v >= s and v < e]

    def project(self, hidden, vocab=None):
        """reference model.py:141-193.  Segmented modes return columns
        segment-major, then in order of appearance in ``vocab``, while the
        bias is added in ``vocab`` order (model.py:152-158,168-179)."""
        w = self.weights
        if self.share_embedding:
            if self.config["D_softmax"]:
                t = np.dot(hidden, w["PM"])
                ys, c0 = [], 0
                for i, (size, s, e) in enumerate(self.segs):
                    blk = self.blocks[i][self._seg_columns(vocab, s, e)] if vocab else self.blocks[i]
                    ys.append(np.dot(t[:, c0:c0 + size], blk.T))
                    c0 += size
                y = np.concatenate(ys, axis=1) + (w["b2"][vocab] if vocab else w["b2"])
            elif self.config["V_table"]:
                t = np.dot(hidden, w["PM"])
                ys = []
                for i, (size, s, e) in enumerate(self.segs):
                    blk = self.blocks[i][self._seg_columns(vocab, s, e)] if vocab else self.blocks[i]
                    ti = t if i == 0 else np.dot(t, self.v_tables[i].T)
                    ys.append(np.dot(ti, blk.T))
                y = np.concatenate(ys, axis=1) + (w["b2"][vocab] if vocab else w["b2"])
            else:
                t = np.dot(hidden, w["PM"])
                if vocab:
                    y = np.dot(t, w["LM"][vocab].T) + w["b2"][vocab]
                else:
                    y = np.dot(t, w["LM"].T) + w["b2"]
        else:
            if vocab:
                # model.py:189 -- indexes ROWS of UM[H, V]; a latent reference
                # bug kept as is (only meaningful with vocab=None).
                y = np.dot(hidden, w["UM"][vocab]) + w["b2"][vocab]
            else:
                y = np.dot(hidden, w["UM"]) + w["b2"]
        return y

    def predict(self, index, hidden, cell, vocab=None):
        """reference model.py:106-123,195-198 with explicit state.
        -> (pred, y, hidden', cell', t_lstm, t_softmax)"""
        t0 = time.time()
        hidden, cell = self.lstm_cell(index, hidden, cell)
        t1 = time.time()
        y = self.project(hidden, vocab)
        pred = np.exp(y) if self.config["self_norm"] else softmax(y)
        t2 = time.time()
        return pred, y, hidden, cell, t1 - t0, t2 - t1


# ---------------------------------------------------------------------- lattice
def build_lattice(text, lexicon, reading_dict, w2i):
    """reference decoder.py:79-135.  -> ends[f] = [(start, reading_len,
    word_idx, word)] for f in 0..len(text); frame 0 holds the <eos> node with
    start -1.  Node order inside a frame is generation order (start ascending,
    then sorted lexicon id), which is the beam's tie-break order."""
    L = len(text)
    ends = [[] for _ in range(L + 1)]
    ends[0].append((-1, 1, w2i["<eos>"], "<eos>"))
    for i in range(L):
        for j in range(L - i):
            sub = text[i:i + j + 1]
            if sub in reading_dict:
                for lex_id in sorted(reading_dict[sub]):
                    word = lexicon[lex_id][0]
                    if word not in w2i:
                        continue                      # OOV skipped, decoder.py:99-103
                    ends[i + j + 1].append((i, j + 1, w2i[word], word))
            if len(ends[i + 1]) == 0:                 # decoder.py:128-130
                ends[i + 1].append((i, 1, w2i["<unk>"], text[i]))
    return ends


def static_vocab(ends, samples=0, top_sampling=False, random_sampling=False, vocab_len=None):
    """reference decoder.py:137-151 (np.random is the caller's to seed)."""
    vocab = sorted(set(n[2] for nodes in ends for n in nodes))
    if samples:
        if random_sampling:
            vocab += [x for x in np.random.randint(vocab_len, size=samples)]
        elif top_sampling:
            vocab += [x for x in range(samples)]
        vocab = sorted(set(vocab))
    return vocab


def dynamic_vocab(ends, samples=0, top_sampling=False, random_sampling=False, vocab_len=None):
    """reference decoder_dynamic.py:30-46: cumulative per-frame vocab lists;
    frame 0's list is NOT de-duplicated when samples are added."""
    lv = {0: sorted(n[2] for n in ends[0])}
    if samples:
        if random_sampling:
            lv[0] += [x for x in np.random.randint(vocab_len, size=samples)]
        elif top_sampling:
            lv[0] += [x for x in range(samples)]
    for i in range(1, len(ends)):
        lv[i] = sorted(set(lv[i - 1]) | set(n[2] for n in ends[i]))
    return lv


# ---------------------------------------------------------------- static decode
class _Frame:
    __slots__ = ("score", "prev", "node", "state", "cell", "prob", "logits")

    def __init__(self):
        self.score, self.prev, self.node = [], [], []
        self.state = self.cell = self.prob = self.logits = None


def _words_of(frames, ends, f, k):
    out = []
    while f >= 0:
        fr = frames[f]
        node = ends[f][fr.node[k]]
        out.append(node[3])
        f, k = fr.prev[k]
    out.reverse()
    return [w for w in out if w != "<eos>"]


def static_decode(lm, ends, beam_width=10, topN=10, vocab=None, perf=None, trace=None):
    """reference decoder.py:164-241.  ``vocab`` = the selected vocab list or
    None.  -> [(neg_log_prob, [word, ...])][:topN].  ``trace`` (a list) gets
    one (scores, prevs, node_ids) tuple per frame."""
    L = len(ends) - 1
    frames = []
    for i in range(L + 1):
        fr = _Frame()
        if i == 0:
            cands = [(0.0, (-1, -1), 0)]
        else:
            cands = []
            for ni, (start, _ln, widx, _w) in enumerate(ends[i]):
                pf = frames[start]
                col = vocab.index(widx) if vocab else widx
                for k in range(len(pf.score)):
                    s = pf.score[k] + (-math.log(pf.prob[k][col]))   # decoder.py:43-49
                    cands.append((s, (start, k), ni))
        if beam_width is not None:
            cands.sort(key=lambda c: c[0])          # stable, decoder.py:227-229
            cands = cands[:beam_width]
        fr.score = [c[0] for c in cands]
        fr.prev = [c[1] for c in cands]
        fr.node = [c[2] for c in cands]
        if i == 0:
            h, c = lm.zero_state(1)
        else:
            h = np.concatenate([frames[pf].state[pk][None] for pf, pk in fr.prev], axis=0)
            c = np.concatenate([frames[pf].cell[pk][None] for pf, pk in fr.prev], axis=0)
        idx = [ends[i][n][2] for n in fr.node]
        pred, y, h, c, t1, t2 = lm.predict(idx, h, c, vocab)          # decoder.py:202-218
        if perf is not None:
            perf[0].append(t1)
            perf[1].append(t2)
        fr.state, fr.cell, fr.prob, fr.logits = h, c, pred, y
        frames.append(fr)
        if trace is not None:
            trace.append((list(fr.score), list(fr.prev), list(fr.node)))
    last = frames[L]
    out = [(last.score[k], _words_of(frames, ends, L, k)) for k in range(len(last.score))]
    return out[:topN]


# --------------------------------------------------------------- dynamic decode
def dynamic_decode(lm, ends, lv, beam_width=10, topN=10, perf=None, trace=None):
    """reference decoder_dynamic.py:49-194.  ``lv`` = dynamic_vocab(...) (it
    is extended in place exactly as the reference extends
    ``self.lattice_vocab``)."""
    L = len(ends) - 1
    self_norm = lm.config["self_norm"]
    frames = []
    root = _Frame()
    root.score, root.prev, root.node = [0.0], [(-1, -1)], [0]
    root.state, root.cell = lm.zero_state(1)
    root.logits, root.prob = [None], [None]
    frames.append(root)
    if trace is not None:
        trace.append(([0.0], [(-1, -1)], [0]))
    for i in range(1, L + 1):
        # ---- _incremental_decode (decoder_dynamic.py:93-148)
        to_fix, missing, fdv = [], set(), {}
        for k in range(i):
            diff = sorted(set(lv[i]) - set(lv[k]))
            if len(diff):
                if k != i - 1:
                    to_fix += [(k, s) for s in range(len(frames[k].score))]
                fdv[k] = diff
                lv[k] += diff
                missing |= set(diff)
        pf = frames[i - 1]
        idx = [ends[i - 1][n][2] for n in pf.node]
        pred, y, h, c, t1, t2 = lm.predict(idx, pf.state, pf.cell, lv[i - 1])
        if perf is not None:
            perf[0].append(t1)
            perf[1].append(t2)
        pf.state, pf.cell = h, c
        pf.prob = [pred[r][None, :] for r in range(pred.shape[0])]
        pf.logits = [y[r] for r in range(y.shape[0])]
        if len(to_fix):
            mv = sorted(missing)
            t0 = time.time()
            lg = lm.project(np.concatenate([frames[k].state[s][None] for k, s in to_fix], axis=0), mv)
            if perf is not None:
                perf[2].append(time.time() - t0)
            dv = fdv[0]        # Path.frame_idx is always 0 (decoder.py:40, decoder_dynamic.py:142)
            cols = [mv.index(x) for x in dv]
            for r, (k, s) in enumerate(to_fix):
                fk = frames[k]
                fk.logits[s] = np.concatenate((fk.logits[s], lg[r][cols]))
                fk.prob[s] = np.exp(fk.logits[s])[None, :] if self_norm else softmax(fk.logits[s])
        # ---- connect + re-score from head (decoder_dynamic.py:69-91,150-175)
        cands = []
        for ni, (start, _ln, widx, _w) in enumerate(ends[i]):
            sf = frames[start]
            col = lv[start].index(widx)
            for k in range(len(sf.score)):
                if self_norm:
                    s = sf.score[k] + (-math.log(sf.prob[k][0][col]))
                else:
                    chain = [(start, k)]
                    while chain[-1][0] > 0:
                        f, q = chain[-1]
                        chain.append(frames[f].prev[q])
                    chain.reverse()
                    for (f0, q0), (f1, q1) in zip(chain[:-1], chain[1:]):
                        n1 = ends[f1][frames[f1].node[q1]]
                        c1 = lv[n1[0]].index(n1[2])
                        frames[f1].score[q1] = frames[f0].score[q0] + (-math.log(frames[f0].prob[q0][0][c1]))
                    s = sf.score[k] + (-math.log(sf.prob[k][0][col]))
                cands.append((s, (start, k), ni))
        if beam_width is not None:
            cands.sort(key=lambda c: c[0])
            cands = cands[:beam_width]
        fr = _Frame()
        fr.score = [c[0] for c in cands]
        fr.prev = [c[1] for c in cands]
        fr.node = [c[2] for c in cands]
        fr.state = np.concatenate([frames[f].state[q][None] for f, q in fr.prev], axis=0)
        fr.cell = np.concatenate([frames[f].cell[q][None] for f, q in fr.prev], axis=0)
        fr.logits = [None] * len(cands)
        fr.prob = [None] * len(cands)
        frames.append(fr)
        if trace is not None:
            trace.append((list(fr.score), list(fr.prev), list(fr.node)))
    last = frames[L]
    out = [(last.score[k], _words_of(frames, ends, L, k)) for k in range(len(last.score))]
    return out[:topN]


# ------------------------------------------------------- character-model decode
def build_char_lattice(text, lexicon, reading_dict, words, c2i):
    """reference decoder.py:79-135 with config['char_rnn'] (:105-124): the word lattice, a node per distinct DISPLAY string of a
    (start, reading) -- at most 201 of them, in sorted lexicon-id order -- indexed by its first character.
    -> ends[f] = [(start, reading_len, first_char_idx, display)]"""
    L = len(text)
    ends = [[] for _ in range(L + 1)]
    ends[0].append((-1, 1, c2i["<eos>"], "<eos>"))
    for i in range(L):
        for j in range(L - i):
            sub = text[i:i + j + 1]
            if sub in reading_dict:
                seen = set()
                for lex_id in sorted(reading_dict[sub]):
                    word = lexicon[lex_id][0]
                    if word not in words:
                        continue                                   # decoder.py:99-103
                    disp = word.split("/")[0]
                    if sum(c not in c2i for c in disp):            # _char_check_oov, decoder.py:266-267
                        continue
                    if disp in seen or len(seen) > 200:            # decoder.py:116-120
                        continue
                    seen.add(disp)
                    ends[i + j + 1].append((i, j + 1, c2i[disp[0]], disp))
            if len(ends[i + 1]) == 0:                              # decoder.py:128-130
                ends[i + 1].append((i, 1, c2i["<unk>"], text[i]))
    return ends


class _CharPath:
    __slots__ = ("score", "words", "text", "h", "c", "prob", "idx", "step", "start")


def _char_word_length(word):
    """decoder.py:269-273"""
    return 1 if word in ("<eos>", "<unk>") else len(word)


def char_decode(lm, ends, c2i, beam_width=10, topN=10, perf=None, trace=None):
    """reference decoder.py:276-341: a frame's candidates are (node, kept path of the node's start frame) pairs whose concatenated
    display strings are new (FIRST occurrence wins, :289-294), scored with the first character from the path's stored distribution
    (:285-287, Path.append_node :43-49); words of several characters then run one LSTM step + softmax per further character, all
    candidates still short of their word's end batched together (_eval_frame, :300-320); stable sort, cut (:330-332); one more step
    on the last character of the survivors (:334).  ``trace``: per frame (scores, (start, word_idx, #nodes))."""
    L = len(ends) - 1
    frames = []

    def step(batch):
        h = np.concatenate([p.h for p in batch], axis=0)
        c = np.concatenate([p.c for p in batch], axis=0)
        pred, _y, h, c, t1, t2 = lm.predict([p.idx for p in batch], h, c, None)      # _batch_predict, decoder.py:202-218
        if perf is not None:
            perf[0].append(t1)
            perf[1].append(t2)
        for k, p in enumerate(batch):
            p.h, p.c, p.prob = h[k][None], c[k][None], pred[k]

    for i in range(L + 1):
        paths = []
        if i == 0:
            p = _CharPath()
            p.score, p.words, p.text, p.prob, p.idx, p.step, p.start = 0.0, ["<eos>"], "<eos>", None, ends[0][0][2], 0, -1
            p.h, p.c = lm.zero_state(1)
            paths.append(p)
        else:
            seen = set()
            for (start, _ln, idx, word) in ends[i]:
                for pp in frames[start]:
                    text = pp.text + word
                    if text in seen:
                        continue
                    seen.add(text)
                    p = _CharPath()
                    p.score = pp.score + (-math.log(pp.prob[idx]))
                    p.words, p.text, p.h, p.c, p.prob, p.idx, p.step, p.start = pp.words + [word], text, pp.h, pp.c, pp.prob, idx, 0, start
                    paths.append(p)
        batch = [p for p in paths if p.step + 1 < _char_word_length(p.words[-1])]
        while batch:                                               # the recursion of _eval_frame, :300-320
            step(batch)
            for p in batch:
                p.step += 1
                p.idx = c2i[p.words[-1][p.step]]
                p.score += -np.log(p.prob[p.idx])
            batch = [p for p in batch if p.step + 1 < _char_word_length(p.words[-1])]
        if beam_width is not None:
            paths.sort(key=lambda p: p.score)
            paths = paths[:beam_width]
        step(paths)
        frames.append(paths)
        if trace is not None:
            trace.append(([p.score for p in paths], [(p.start, p.idx, len(p.words)) for p in paths]))
    return [(p.score, [w for w in p.words if w != "<eos>"]) for p in frames[L]][:topN]


# ------------------------------------------------------------- file-level API
class OracleDecoder:
    """File-loading wrapper with the reference's signatures (decoder.py:54-77,
    220-241).  ``root`` replaces the reference's frozen config.root_path."""

    dynamic = False

    def __init__(self, root, experiment_id=0, comp=0):
        exp = os.path.join(root, "train", "experiments", str(experiment_id))
        with open(os.path.join(exp, "config.json"), "rt") as f:
            self.config = json.loads(f.read())
        wfile = "lstm_weights_comp_%d.pkl" % comp if comp else "lstm_weights.pkl"   # model.py:74-78
        with open(os.path.join(exp, "weights", wfile), "rb") as f:
            weights = pickle.load(f)
        with open(os.path.join(root, "data", "lexicon.pkl"), "rb") as f:
            self.full_lexicon = pickle.load(f)
        with open(os.path.join(root, "data", "reading_dict.pkl"), "rb") as f:
            self.full_reading_dict = pickle.load(f)
        lex = [("<unk>", 0)] + self.full_lexicon[: self.config["vocab_size"] - 1]   # train/data.py:17-22
        self.w2i = {x[0]: i for i, x in enumerate(lex)}
        self.i2w = {v: k for k, v in self.w2i.items()}
        self.model = OracleLM(self.config, weights)
        self.lattice_vocab = None
        self.perf_sen = 0
        self.perf_log_lstm, self.perf_log_softmax = [], []
        self.perf_log_fix_vocab, self.perf_log_fix_lattice_path_prob = [], []
        self.last_trace = None

    def _check_oov(self, word):
        return word not in self.w2i

    def decode(self, input, topN=10, beam_width=10, vocab_select=False, samples=0,
               top_sampling=False, random_sampling=False):
        ends = build_lattice(input, self.full_lexicon, self.full_reading_dict, self.w2i)
        self.backward_lookup = ends
        perf = (self.perf_log_lstm, self.perf_log_softmax, self.perf_log_fix_vocab)
        self.last_trace = []
        if self.dynamic:
            if not vocab_select:
                raise TypeError("'NoneType' object is not subscriptable")   # decoder_dynamic.py:114
            self.lattice_vocab = dynamic_vocab(ends, samples, top_sampling, random_sampling, len(self.w2i))
            out = dynamic_decode(self.model, ends, self.lattice_vocab, beam_width, topN, perf, self.last_trace)
        else:
            if vocab_select:
                self.lattice_vocab = static_vocab(ends, samples, top_sampling, random_sampling, len(self.w2i))
            # the reference keeps a stale lattice_vocab across calls (decoder.py:62,176)
            out = static_decode(self.model, ends, beam_width, topN, self.lattice_vocab, perf, self.last_trace)
        self.perf_sen += 1
        return out


class OracleDynamicDecoder(OracleDecoder):
    dynamic = True


class OracleCharRNNDecoder(OracleDecoder):
    """reference decoder.py:244-341 as evidently intended (module docstring: parity unpinned): the word index decides what is in
    the vocabulary, CharVocab's character index (train/data.py:28-47) what the model steps over."""

    def __init__(self, root, experiment_id=0, comp=0):
        super(OracleCharRNNDecoder, self).__init__(root, experiment_id, comp)
        self.words = self.w2i
        lex = [("<unk>", 0)] + self.full_lexicon[: self.config["vocab_size"] - 1]
        c2i = {"<unk>": 0, "<eos>": 1}
        for item in lex[2:]:
            for ch in item[0].split("/")[0]:
                if ch not in c2i:
                    c2i[ch] = len(c2i)
        self.w2i = c2i
        self.i2w = {v: k for k, v in c2i.items()}

    def _check_oov(self, word):
        return word not in self.words

    def decode(self, input, topN=10, beam_width=10, vocab_select=False, samples=0,
               top_sampling=False, random_sampling=False):
        ends = build_char_lattice(input, self.full_lexicon, self.full_reading_dict, self.words, self.w2i)
        self.backward_lookup = ends
        if vocab_select:                # decoder.py:132-133: the list is built and never read by this class
            self.lattice_vocab = static_vocab(ends, samples, top_sampling, random_sampling, len(self.w2i))
        self.last_trace = []
        out = char_decode(self.model, ends, self.w2i, beam_width, topN, (self.perf_log_lstm, self.perf_log_softmax), self.last_trace)
        self.perf_sen += 1
        return out
