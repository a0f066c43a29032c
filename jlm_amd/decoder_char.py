"""Character-model decoder on the word lattice: counterpart of ``CharRNNDecoder`` in the reference
(decoder/decoder.py:244-341), as evidently intended.

The reference's class cannot run as shipped: ``_check_oov`` reads ``self.vocab.words`` (decoder.py:263-264), which no
``Vocab`` defines (train/data.py:15-47), and ``Decoder._load_vocab`` (decoder.py:70-73) hands it the WORD index where
its character steps (``self.w2i[word[0]]``, decoder.py:123; ``self.w2i[n.word[n.char_rnn_step]]``, :316) and
``_char_check_oov`` (:266-267) need ``CharVocab.c2i``.  Wired the only way those statements make sense -- ``vocab`` a
``CharVocab``, ``vocab.words`` its word index, ``w2i`` / ``i2w`` its character index -- the class runs, and that run
(one method supplied by a subclass at run time inside ``tools/make_golden.py``, nothing edited) is what the golden vectors
of this decoder hold (``tests/golden/char.json``); PARITY UNPINNED in the sense of the task's rules.

What the search does (decoder.py:276-341):

  * the lattice is the WORD lattice with one node per distinct display string of a (start, reading) -- at most 201, in
    sorted lexicon-id order -- indexed by its FIRST character (decoder.py:105-124);
  * a frame's candidates are (node, kept path of the node's start frame) pairs whose concatenated display strings have
    not been seen in this frame -- the FIRST pair wins, not the best (:289-294); each starts with the first character's
    probability under the path's stored distribution (:285-287, Path.append_node :43-49);
  * words of several characters then take one LSTM step + softmax per further character, every candidate still short of
    its word's end in ONE batch per character position (``_eval_frame``, :300-320);
  * stable sort by score, cut to the beam (:330-332), one more step on the last character of the survivors (:334).

MI355X form: ``decode_batch`` runs the frames of a whole batch of sentences in lock step, so that every one of those
batches holds the candidates of ALL sentences; hidden and cell rows never leave HBM (one pool per batch: the kept paths'
rows, then the frame's scratch rows, written in place by ``jlm_lstm_step`` through row slices), the softmax rows stay on
the device and only the (row, character) entries the search asks for come back (one gather per character position).  The
search itself -- string dedup, stable sort -- is host logic, as in the reference.  The kernels are those behind
``LSTM_Model.predict`` (``LSTM_Model.step_resident``): a character vocabulary is a few thousand rows, nothing here needs
the fused normaliser of the word models.
"""
import math
import os
import pickle

import numpy as np

from . import config as _config
from .data import CharVocab
from .decoder import Decoder, Node
from .model import LSTM_Model

_P_FLOOR = 1.0e-45       # a float32 softmax entry that underflowed to 0 (a logit 103 below the row's largest) is scored as this


class _Path(object):
    __slots__ = ("score", "prev", "word", "text", "row", "pf", "pk", "idx", "step", "start", "n_nodes")


def _word_length(word):
    """decoder.py:269-273"""
    return 1 if word in ("<eos>", "<unk>") else len(word)


class CharRNNDecoder(Decoder):
    """``CharRNNDecoder(experiment_id=0, comp=0)`` / ``.decode(input, topN=10, beam_width=10, vocab_select=False, samples=0,
    top_sampling=False, random_sampling=False) -> [(neg_log_prob, [display string, ...])]`` (decoder.py:244-341) plus
    ``decode_batch``.  ``beam_width=None`` keeps every candidate (:330: exponential; ``max_unpruned_paths`` bounds a frame)."""

    def __init__(self, experiment_id=0, comp=0, device=None):
        self.config = _config.load_config_dict(experiment_id)
        self._load_vocab()
        with open(os.path.join(_config.root_path, 'data', 'lexicon.pkl'), 'rb') as f:
            self.full_lexicon = pickle.load(f)
        with open(os.path.join(_config.root_path, 'data', 'reading_dict.pkl'), 'rb') as f:
            self.full_reading_dict = pickle.load(f)
        self.model = LSTM_Model(experiment_id, comp, device=device)
        if self.model.dev.V < len(self.w2i):
            raise ValueError("the model's softmax has %d rows, the character index %d entries (is this a character model? "
                             "config['char_rnn'] = %r)" % (self.model.dev.V, len(self.w2i), self.config.get('char_rnn')))
        self.lattice_vocab = None
        self.backward_lookup = None
        self.perf_sen = 0
        self.perf_log_lstm = []
        self.perf_log_softmax = []
        self.perf_timing = False
        self.compat_quirks = False
        self.max_unpruned_paths = 20000
        self.max_batch = 256             # sentences per lock-step batch
        self._nodes_of = {}              # reading -> [(first character's index, display string)] (decoder.py:93-124), filled on demand
        print('Char RNN decoder loaded')

    # ---------------------------------------------------------------- vocabulary, lattice (host logic)
    def _load_vocab(self):
        self.vocab = CharVocab(self.config['vocab_size'])
        self.vocab.words = self.vocab.w2i        # what decoder.py:264 reads
        self.w2i = self.vocab.c2i
        self.i2w = self.vocab.i2c

    def _check_oov(self, word):
        return word not in self.vocab.words

    def _char_check_oov(self, word):
        return sum([c not in self.w2i for c in word.split('/')[0]])

    def _word_length(self, word):
        return _word_length(word)

    def _reading_nodes(self, reading):
        nodes = self._nodes_of.get(reading)
        if nodes is None:
            nodes, seen = [], set()
            for lexicon_id in sorted(self.full_reading_dict[reading]):
                word = self.full_lexicon[lexicon_id][0]
                if self._check_oov(word) or self._char_check_oov(word):
                    continue
                disp = word.split('/')[0]
                if disp in seen or len(seen) > 200:
                    continue
                seen.add(disp)
                nodes.append((self.w2i[disp[0]], disp))
            self._nodes_of[reading] = nodes
        return nodes

    def _ends(self, input):
        """ends[f] = [(start, reading length, first character's index, display string)] of the nodes ending at frame f"""
        L = len(input)
        ends = [[] for _ in range(L + 1)]
        ends[0].append((-1, 1, self.w2i['<eos>'], '<eos>'))
        rd = self.full_reading_dict
        for i in range(L):
            for j in range(L - i):
                sub = input[i:i + j + 1]
                if sub in rd:
                    tgt = ends[i + j + 1]
                    for idx, disp in self._reading_nodes(sub):
                        tgt.append((i, j + 1, idx, disp))
                if j == 0 and not ends[i + 1]:
                    ends[i + 1].append((i, 1, self.w2i['<unk>'], input[i]))
        return ends

    def _build_lattice(self, input, vocab_select=False, samples=0, top_sampling=False, random_sampling=False):
        """dict frame -> [Node], the reference's shape (decoder.py:79-135)"""
        ends = self._ends(input)
        if vocab_select:
            self._build_lattice_vocab(ends, samples, top_sampling, random_sampling)
        return {f: [Node(s, l, w, word) for (s, l, w, word) in nodes] for f, nodes in enumerate(ends)}

    def _build_lattice_vocab(self, ends, samples=0, top_sampling=False, random_sampling=False):
        """decoder.py:137-151: built when asked for and never read by this class (its frames index the full softmax)"""
        lv = sorted(set(n[2] for nodes in ends for n in nodes))
        if samples:
            if random_sampling:
                lv += [int(x) for x in np.random.randint(len(self.w2i), size=samples)]
            elif top_sampling:
                lv += list(range(samples))
            lv = sorted(set(lv))
        self.lattice_vocab = lv

    # ---------------------------------------------------------------- search
    def decode(self, input, topN=10, beam_width=10, vocab_select=False, samples=0, top_sampling=False, random_sampling=False):
        out = self.decode_batch([input], topN, beam_width, vocab_select, samples, top_sampling, random_sampling)[0]
        self.backward_lookup = {f: [Node(s, l, w, word) for (s, l, w, word) in nodes] for f, nodes in enumerate(self._last_ends)}
        return out

    def decode_batch(self, inputs, topN=10, beam_width=10, vocab_select=False, samples=0, top_sampling=False,
                     random_sampling=False):
        inputs = list(inputs)
        if beam_width is not None and int(beam_width) < 1:
            raise ValueError("beam_width must be at least 1 (or None)")
        out = []
        for b0 in range(0, len(inputs), self.max_batch):
            out += self._decode_lockstep(inputs[b0:b0 + self.max_batch], topN, beam_width, vocab_select, samples, top_sampling,
                                         random_sampling)
        self.perf_sen += len(inputs)
        return out

    def _decode_lockstep(self, inputs, topN, beam_width, vocab_select, samples, top_sampling, random_sampling):
        m = self.model
        d = m.dev
        torch = d.torch
        dev = m.device
        S = len(inputs)
        ends = [self._ends(x) for x in inputs]
        self._last_ends = ends[-1]
        if vocab_select:
            for e in ends:                       # (np.random is drawn per sentence, in order, as decode() after decode() would)
                self._build_lattice_vocab(e, samples, top_sampling, random_sampling)
        H = d.H
        beam = None if beam_width is None else int(beam_width)
        Lmax = max(len(x) for x in inputs)
        # state pool: row 0 is the zero state; kept paths' rows follow frame by frame, the frame's scratch rows behind them
        cap = 1 + (S * (beam if beam else 64) * 4)
        pool = {"h": torch.zeros((cap, H), device=dev, dtype=torch.float32), "c": torch.zeros((cap, H), device=dev, dtype=torch.float32)}
        top = 1

        def reserve(rows):
            cur = pool["h"].shape[0]
            if rows <= cur:
                return
            new = max(rows, 2 * cur)
            for k in ("h", "c"):
                t = torch.zeros((new, H), device=dev, dtype=torch.float32)
                t[:cur] = pool[k]
                pool[k] = t

        def ints(a):
            return torch.as_tensor(np.asarray(a, dtype=np.int32), device=dev)

        def longs(a):
            return torch.as_tensor(np.asarray(a, dtype=np.int64), device=dev)

        def log_perf(t):
            if t is not None:
                self.perf_log_lstm.append(t[0])
                self.perf_log_softmax.append(t[1])

        P = {}                                   # frame -> softmax rows of its kept paths (all sentences), on the device
        frames = [[None] * (len(x) + 1) for x in inputs]
        for i in range(Lmax + 1):
            active = [s for s in range(S) if len(inputs[s]) >= i]
            cands = {}
            if i == 0:
                for s in active:
                    p = _Path()
                    p.score, p.prev, p.word, p.text, p.row, p.idx, p.step, p.start, p.n_nodes = 0.0, None, '<eos>', '<eos>', 0, \
                        ends[s][0][0][2], 0, -1, 1
                    cands[s] = [p]
            else:
                ask = {}                         # start frame -> ([row of P[frame]], [character], [candidate])
                for s in active:
                    lst, seen = [], set()
                    for (start, _ln, idx, word) in ends[s][i]:
                        for pp in frames[s][start]:
                            text = pp.text + word
                            if text in seen:
                                continue
                            seen.add(text)
                            p = _Path()
                            p.score, p.prev, p.word, p.text, p.row, p.idx, p.step, p.start, p.n_nodes = pp.score, pp, word, text, pp.row, \
                                idx, 0, start, pp.n_nodes + 1
                            a = ask.setdefault(pp.pf, ([], [], []))
                            a[0].append(pp.pk)
                            a[1].append(idx)
                            a[2].append(p)
                            lst.append(p)
                    if beam is None and len(lst) > self.max_unpruned_paths:
                        raise ValueError("beam_width=None: frame %d holds %d hypotheses (max_unpruned_paths = %d)" % (
                            i, len(lst), self.max_unpruned_paths))
                    cands[s] = lst
                for pf, (rows, cols, ps) in ask.items():
                    pr = P[pf][longs(rows), longs(cols)].double().cpu().numpy()
                    for p, v in zip(ps, pr):
                        p.score += -math.log(max(float(v), _P_FLOOR))              # Path.append_node, decoder.py:43-49
            # words of several characters: one step + softmax per further character, all sentences' candidates together
            batch = [p for s in active for p in cands[s] if _word_length(p.word) > 1]
            base = top
            while batch:
                n = len(batch)
                reserve(base + n)
                pred, _nc, t = m.step_resident(pool["h"], pool["c"], ints([p.row for p in batch]), ints([p.idx for p in batch]),
                                               pool["h"][base:base + n], pool["c"][base:base + n], self.perf_timing)
                log_perf(t)
                nxt = [self.w2i[p.word[p.step + 1]] for p in batch]
                pr = pred[longs(np.arange(n)), longs(nxt)].double().cpu().numpy()
                for k, p in enumerate(batch):
                    p.row, p.step, p.idx = base + k, p.step + 1, nxt[k]
                    p.score += -np.log(max(float(pr[k]), _P_FLOOR))               # decoder.py:317
                base += n
                batch = [p for p in batch if p.step + 1 < len(p.word)]
            for s in active:
                lst = cands[s]
                if beam is not None:
                    lst.sort(key=lambda p: p.score)                               # stable, decoder.py:331
                    lst = lst[:beam]
                frames[s][i] = lst
            if i == Lmax:
                break                            # (the reference steps the last frame too and drops the result, decoder.py:334-336)
            step_paths = [p for s in active if len(inputs[s]) > i for p in frames[s][i]]
            n = len(step_paths)
            h_new = torch.empty((n, H), device=dev, dtype=torch.float32)
            c_new = torch.empty((n, H), device=dev, dtype=torch.float32)
            pred, _nc, t = m.step_resident(pool["h"], pool["c"], ints([p.row for p in step_paths]), ints([p.idx for p in step_paths]),
                                           h_new, c_new, self.perf_timing)
            log_perf(t)
            reserve(top + n)
            pool["h"][top:top + n] = h_new
            pool["c"][top:top + n] = c_new
            P[i] = pred
            for k, p in enumerate(step_paths):
                p.row, p.pf, p.pk = top + k, i, k
            top += n
            # (softmax rows of frames no later node can start in could be dropped here; a batch's worth is a few hundred MB at most)
        out = []
        for s in range(S):
            res = []
            for p in frames[s][len(inputs[s])][:topN]:
                words, q = [], p
                while q is not None:
                    if q.word != '<eos>':
                        words.append(q.word)
                    q = q.prev
                words.reverse()
                res.append((p.score, words))
            out.append(res)
        self._last_frames = frames
        return out
