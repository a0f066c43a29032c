"""Host-side word lattice over kana input, packed as CSR for the device.

Follows ``Decoder._build_lattice`` / ``_build_lattice_vocab`` of the reference
(decoder/decoder.py:79-151) and ``DynamicDecoder._build_lattice_vocab``
(decoder/decoder_dynamic.py:30-46), for a whole batch of sentences at once.

Node order inside a (frame, sentence) cell is the reference's generation order
(start position ascending, then sorted lexicon id): it is the beam's tie-break
order, so it is part of the contract.

Arrays (int32, frame-major cells ``cell = frame * n_sent + sentence``):

  node_start[n], node_word[n]   start frame (-1 for <eos>) / softmax row
  node_lex[n]                   lexicon index (>=0), -1 = <eos>, -2 = raw-kana <unk> fallback
  end_off[cell .. cell+1]       nodes ENDING in the cell (node ids are in this order)
  sg_off / sg_word / sg_node    nodes STARTING in the cell (edge-logit work lists)
"""
import numpy as np


class LatticeBuilder:
    """Reading dictionary pre-filtered to in-vocabulary words (the reference
    re-checks OOV per lookup, decoder.py:96-103; the result is the same)."""

    def __init__(self, lexicon, reading_dict, w2i):
        self.w2i = w2i
        self.eos = w2i['<eos>']
        self.unk = w2i['<unk>']
        self.lexicon = lexicon
        self.lex_words = np.array([x[0] for x in lexicon], dtype=object)
        self.table = {}
        self.max_len = 1
        for reading, ids in reading_dict.items():
            words, lex = [], []
            for lid in sorted(ids):
                w = lexicon[lid][0]
                wi = w2i.get(w)
                if wi is None:
                    continue
                words.append(wi)
                lex.append(lid)
            if words:
                self.table[reading] = (words, lex)
                if len(reading) > self.max_len:
                    self.max_len = len(reading)

    def sentence_nodes(self, text):
        """-> (end, start, word, lex) python lists in generation order."""
        L = len(text)
        end, start, word, lex = [0], [-1], [self.eos], [-1]
        has = [False] * (L + 1)
        table, max_len = self.table, self.max_len
        for i in range(L):
            for j in range(min(L - i, max_len)):
                hit = table.get(text[i:i + j + 1])
                if hit is not None:
                    n = len(hit[0])
                    end.extend([i + j + 1] * n)
                    start.extend([i] * n)
                    word.extend(hit[0])
                    lex.extend(hit[1])
                    has[i + j + 1] = True
                if j == 0 and not has[i + 1]:        # decoder.py:128-130
                    end.append(i + 1)
                    start.append(i)
                    word.append(self.unk)
                    lex.append(-2)
                    has[i + 1] = True
        return end, start, word, lex


class BatchLattice:
    def __init__(self, builder, texts, beam):
        self.texts = list(texts)
        self.builder = builder
        B = len(self.texts)
        self.n_sent = B
        self.beam = int(beam)
        self.sent_len = np.array([len(t) for t in self.texts], dtype=np.int32)
        self.n_frames = int(self.sent_len.max()) + 1 if B else 1
        ends, starts, words, lexs, sents = [], [], [], [], []
        for s, t in enumerate(self.texts):
            e, st, w, lx = builder.sentence_nodes(t)
            ends.append(np.asarray(e, dtype=np.int64))
            starts.append(np.asarray(st, dtype=np.int32))
            words.append(np.asarray(w, dtype=np.int32))
            lexs.append(np.asarray(lx, dtype=np.int32))
            sents.append(np.full(len(e), s, dtype=np.int64))
        end = np.concatenate(ends)
        start = np.concatenate(starts)
        word = np.concatenate(words)
        lex = np.concatenate(lexs)
        sent = np.concatenate(sents)
        ncell = self.n_frames * B
        cell = end * B + sent
        order = np.argsort(cell, kind="stable")
        self.node_start = np.ascontiguousarray(start[order])
        self.node_word = np.ascontiguousarray(word[order])
        self.node_lex = np.ascontiguousarray(lex[order])
        self.node_sent = np.ascontiguousarray(sent[order].astype(np.int32))
        self.node_end = np.ascontiguousarray(end[order].astype(np.int32))
        self.n_nodes = int(order.shape[0])
        counts = np.bincount(cell, minlength=ncell)
        self.end_off = np.zeros(ncell + 1, dtype=np.int32)
        np.cumsum(counts, out=self.end_off[1:])
        self.max_cands = int(counts.max()) * self.beam
        # nodes grouped by the cell they START in
        ids = np.nonzero(self.node_start >= 0)[0]
        scell = self.node_start[ids].astype(np.int64) * B + self.node_sent[ids]
        so = np.argsort(scell, kind="stable")
        self.sg_node = np.ascontiguousarray(ids[so].astype(np.int32))
        self.sg_word = np.ascontiguousarray(self.node_word[self.sg_node])
        scount = np.bincount(scell, minlength=ncell)
        self.sg_off = np.zeros(ncell + 1, dtype=np.int32)
        np.cumsum(scount, out=self.sg_off[1:])

    # ---- per sentence views (used by vocabulary selection and by the tests)
    def frame_nodes(self, s, f):
        a, b = self.end_off[f * self.n_sent + s], self.end_off[f * self.n_sent + s + 1]
        return range(int(a), int(b))

    def backward_lookup(self, s):
        """The reference's ``backward_lookup`` as plain tuples
        ``(start_idx, reading_length, word_idx, word)`` per frame."""
        out = []
        for f in range(int(self.sent_len[s]) + 1):
            out.append([(int(self.node_start[n]), (f - int(self.node_start[n])) if self.node_start[n] >= 0 else 1,
                         int(self.node_word[n]), self.word_str(n)) for n in self.frame_nodes(s, f)])
        return out

    def word_str(self, n):
        lx = int(self.node_lex[n])
        if lx >= 0:
            return self.builder.lexicon[lx][0]
        if lx == -1:
            return '<eos>'
        return self.texts[int(self.node_sent[n])][int(self.node_start[n])]

    def words_of(self, node_ids):
        """Vectorised node id -> word string (object array)."""
        node_ids = np.asarray(node_ids, dtype=np.int64)
        lx = self.node_lex[node_ids]
        out = self.builder.lex_words[np.maximum(lx, 0)]
        special = np.nonzero(lx < 0)[0]
        if special.size:
            out = out.copy()
            for i in special:
                out[i] = self.word_str(int(node_ids[i]))
        return out

    # ---- vocabulary selection
    def static_vocab(self, samples=0, top_sampling=False, random_sampling=False, vocab_len=None):
        """Per sentence sorted unique word ids over its lattice (+ samples),
        reference decoder.py:137-151.  -> (vs_words, vs_off, python lists)"""
        lists = []
        order = np.argsort(self.node_sent, kind="stable")
        bounds = np.searchsorted(self.node_sent[order], np.arange(self.n_sent + 1))
        for s in range(self.n_sent):
            v = np.unique(self.node_word[order[bounds[s]:bounds[s + 1]]]).tolist()
            if samples:
                if random_sampling:
                    v += [int(x) for x in np.random.randint(vocab_len, size=samples)]
                elif top_sampling:
                    v += [x for x in range(samples)]
                v = sorted(set(v))
            lists.append(v)
        return _csr(lists) + (lists,)

    def dynamic_vocab(self, samples=0, top_sampling=False, random_sampling=False, vocab_len=None):
        """Cumulative per-frame vocabularies of the incremental decoder
        (decoder_dynamic.py:30-46) as the word lists the device needs:

          init lists  cell (k, s): what frame k's rows are first normalised over
                      = lv[k] + delta[k+1]   (frame 0 keeps duplicated samples)
          delta lists cell (i, s): sorted(set(lv[i]) - set(lv[i-1])), appended to
                      every older frame at step i (decoder_dynamic.py:112-127)

        -> (init_words, init_off, delta_words, delta_off, lv_final) where
        lv_final[s][k] is the reference's ``lattice_vocab[k]`` list after the
        decode (original order + appended deltas)."""
        B, F = self.n_sent, self.n_frames
        init = [[] for _ in range(F * B)]
        delta = [[] for _ in range(F * B)]
        lv_final = []
        for s in range(B):
            L = int(self.sent_len[s])
            fw = [sorted(int(self.node_word[n]) for n in self.frame_nodes(s, f)) for f in range(L + 1)]
            lv = {0: list(fw[0])}
            if samples:
                if random_sampling:
                    lv[0] += [int(x) for x in np.random.randint(vocab_len, size=samples)]
                elif top_sampling:
                    lv[0] += [x for x in range(samples)]
            for i in range(1, L + 1):
                lv[i] = sorted(set(lv[i - 1]) | set(fw[i]))
            d = {i: sorted(set(lv[i]) - set(lv[i - 1])) for i in range(1, L + 1)}
            for i in range(1, L + 1):
                delta[i * B + s] = d[i]
            for k in range(L):
                init[k * B + s] = lv[k] + d[k + 1]
            final = {}
            for k in range(L + 1):
                final[k] = list(lv[k])
                for i in range(k + 1, L + 1):
                    final[k] += d[i]
            lv_final.append(final)
        iw, io = _csr(init)
        dw, do = _csr(delta)
        return iw, io, dw, do, lv_final


def _csr(lists):
    off = np.zeros(len(lists) + 1, dtype=np.int32)
    if lists:
        np.cumsum([len(x) for x in lists], out=off[1:])
    flat = np.fromiter((v for l in lists for v in l), dtype=np.int32, count=int(off[-1]))
    return flat, off
