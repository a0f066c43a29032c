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
import ctypes
import os
import threading

import numpy as np

_HOST_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libjlm_host.so")
_host = None


def host_lib():
    """libjlm_host.so (include/jlm_host.h) or None when it has not been built."""
    global _host
    if _host is None:
        if not os.path.exists(_HOST_LIB):
            _host = False
        else:
            l = ctypes.CDLL(_HOST_LIB)
            P, I, L64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
            l.jlm_host_abi_version.restype = ctypes.c_int
            l.jlm_lexicon_create.argtypes = [P, P, P, P, P, I, I, I]
            l.jlm_lexicon_create.restype = P
            l.jlm_lexicon_destroy.argtypes = [P]
            l.jlm_lattice_build.argtypes = [P, P, P, I, I, L64, P, P, P, P, P, P, P, P, P, P, I]
            l.jlm_lattice_build.restype = L64
            l.jlm_static_vocab.argtypes = [P, P, L64, I, I, L64, P, P, I]
            l.jlm_static_vocab.restype = L64
            l.jlm_static_vocab_cells.argtypes = [P, P, I, I, I, L64, P, P]
            l.jlm_static_vocab_cells.restype = L64
            l.jlm_dynamic_vocab.argtypes = [P, P, P, I, I, P, P, L64, L64, P, P, P, P, P, I]
            l.jlm_dynamic_vocab.restype = L64
            _host = l if l.jlm_host_abi_version() == 3 else False
    return _host or None


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _utf32(s):
    return np.frombuffer(s.encode("utf-32-le"), dtype=np.uint32)


class LatticeBuilder:
    """Reading dictionary pre-filtered to in-vocabulary words (the reference
    re-checks OOV per lookup, decoder.py:96-103; the result is the same)."""

    def __init__(self, lexicon, reading_dict, w2i):
        self.w2i = w2i
        self.eos = w2i['<eos>']
        self.unk = w2i['<unk>']
        self.lexicon = lexicon
        self.lex_list = [x[0] for x in lexicon]
        self.lex_words = np.array(self.lex_list, dtype=object)
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
        self._native = None
        self._native_lock = threading.Lock()          # decode_batch's prefetch threads all ask for the trie on first use
        # threads per build: ONE.  A 256 x 20-kana batch is 1.44 ms single-threaded on the GPU box and 1.8-2.4 ms with 2-16
        # threads (tools/probes/lattice_build_time.py: the walk over a batch is shorter than spawning and joining the threads);
        # decode_batch runs several builds side by side instead.  JLM_LATTICE_THREADS overrides.
        self.n_threads = max(1, int(os.environ.get("JLM_LATTICE_THREADS", "1")))
        self.use_native = os.environ.get("JLM_NATIVE_LATTICE", "1") != "0"

    def native(self):
        """Handle of the C++ trie (built on first use), or None."""
        if not self.use_native or host_lib() is None:
            return None
        if self._native is not None:
            return self._native
        with self._native_lock:
            if self._native is not None:
                return self._native
            readings = list(self.table)
            cps = [_utf32(r) for r in readings]
            r_off = np.zeros(len(readings) + 1, dtype=np.int32)
            np.cumsum([len(c) for c in cps], out=r_off[1:])
            e_off = np.zeros(len(readings) + 1, dtype=np.int32)
            np.cumsum([len(self.table[r][0]) for r in readings], out=e_off[1:])
            self._nat_arrays = (np.concatenate(cps) if cps else np.zeros(1, np.uint32), r_off, e_off,
                                np.fromiter((w for r in readings for w in self.table[r][0]), dtype=np.int32,
                                            count=int(e_off[-1])),
                                np.fromiter((x for r in readings for x in self.table[r][1]), dtype=np.int32,
                                            count=int(e_off[-1])))
            a = self._nat_arrays
            self._native = host_lib().jlm_lexicon_create(_ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]),
                                                         len(readings), self.eos, self.unk)
        return self._native

    def __del__(self):
        try:
            if getattr(self, "_native", None) and host_lib() is not None:
                host_lib().jlm_lexicon_destroy(self._native)
        except Exception:
            pass

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


class StagingPool:
    """Page-locked int32 blocks for the lattice arrays that go to the device (node_start, node_word, sg_word, sg_node): the native
    builder writes them where the H2D copy reads them -- no memcpy into a staging buffer on the enqueueing thread (3.5 MB per
    256-sentence batch).  A block goes back to the pool when its batch has been read out (BatchLattice.release); blocks of lattices
    that stay referenced (Decoder.last_lattice) are simply not returned.  Without a GPU the blocks are plain tensors (same code
    path under the CPU tests)."""

    class Block:
        __slots__ = ("tensor", "np", "n")

        def __init__(self, tensor):
            self.tensor, self.np, self.n = tensor, tensor.numpy(), int(tensor.numel())

    def __init__(self, torch, pinned):
        self.torch, self.pinned = torch, bool(pinned)
        self._free = {}
        self._lock = threading.Lock()
        self.allocated = 0

    def get(self, n_ints):
        n = 1 << max(16, int(n_ints - 1).bit_length())          # sizes in powers of two: a handful of distinct ones per process
        with self._lock:
            lst = self._free.get(n)
            if lst:
                return lst.pop()
            self.allocated += 1
        t = self.torch.empty(n, dtype=self.torch.int32, pin_memory=self.pinned)
        return StagingPool.Block(t)

    def put(self, blk):
        with self._lock:
            self._free.setdefault(blk.n, []).append(blk)


class BatchLattice:
    def __init__(self, builder, texts, beam, pool=None):
        self.texts = list(texts)
        self.builder = builder
        self._pool, self._block, self.block_off = pool, None, None
        B = len(self.texts)
        self.n_sent = B
        self.beam = int(beam)
        self.sent_len = np.array([len(t) for t in self.texts], dtype=np.int32)
        self.n_frames = int(self.sent_len.max()) + 1 if B else 1
        nat = builder.native() if B else None
        if nat is not None:
            self._build_native(nat)
            return
        self._pool = None
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

    def _build_native(self, nat):
        """Same arrays through libjlm_host.so (trie walk, threads over sentences)."""
        lib = host_lib()
        B, F = self.n_sent, self.n_frames
        text = _utf32("".join(self.texts))
        t_off = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(self.sent_len, out=t_off[1:])
        assert int(t_off[-1]) == text.size, "non-BMP / surrogate input is not supported by the native builder"
        ncell = F * B
        self.end_off = np.zeros(ncell + 1, dtype=np.int32)
        self.sg_off = np.zeros(ncell + 1, dtype=np.int32)
        mx = np.zeros(1, dtype=np.int32)
        cap = max(1024, int(text.size) * 48 + B)
        while True:
            if self._pool is not None:
                # the four arrays the device reads at offsets 0, cap, 2 cap, 3 cap of ONE page-locked block; the three host-only ones
                # behind them (a pooled block is touched memory: fresh np.empty arrays cost a page fault per 4 KB on every build)
                blk = self._pool.get(7 * cap)
                dev = [blk.np[i * cap:(i + 1) * cap] for i in range(7)]
                arrs = [dev[0], dev[1], dev[4], dev[5], dev[6], dev[2], dev[3]]
            else:
                blk = None
                arrs = [np.empty(cap, dtype=np.int32) for _ in range(7)]
            n = lib.jlm_lattice_build(nat, _ptr(text), _ptr(t_off), B, F, cap, *[_ptr(a) for a in arrs[:5]],
                                      _ptr(self.end_off), _ptr(self.sg_off), _ptr(arrs[5]), _ptr(arrs[6]), _ptr(mx),
                                      self.builder.n_threads)
            if n <= cap:
                break
            if blk is not None:
                self._pool.put(blk)
            cap = int(n)
        n = int(n)
        if blk is not None:
            self._block = blk
            self.block_off = dict(node_start=0, node_word=cap, sg_node=2 * cap, sg_word=3 * cap)
        self.n_nodes = n
        self.node_start, self.node_word, self.node_lex, self.node_sent, self.node_end = (a[:n] for a in arrs[:5])
        ns = int(self.sg_off[-1])
        self.sg_node, self.sg_word = arrs[5][:ns], arrs[6][:ns]
        self.max_cands = int(mx[0]) * self.beam

    def release(self):
        """Give the page-locked block back (the batch has been read out; node_start / node_word / sg_* are dead from here on)."""
        blk, self._block = self._block, None
        if blk is not None and self._pool is not None:
            self.node_start = self.node_word = self.sg_node = self.sg_word = self.node_lex = self.node_sent = self.node_end = None
            self._pool.put(blk)

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
        lib = host_lib()
        if lib is not None and self.builder.use_native and not (samples and random_sampling):
            top = int(samples) if (samples and top_sampling) else 0
            off = np.zeros(self.n_sent + 1, dtype=np.int32)
            cap = self.n_nodes + (top + 1) * self.n_sent
            words = np.empty(cap, dtype=np.int32)
            n = lib.jlm_static_vocab_cells(_ptr(np.ascontiguousarray(self.node_word)), _ptr(self.end_off), self.n_sent,
                                           self.n_frames, top, cap, _ptr(words), _ptr(off))
            assert n <= cap
            words = words[:int(n)]
            return words, off, _LazyLists(words, off)
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

        lv[] is cumulative, so every init list is a slice of one per-sentence sequence
        (include/jlm_host.h, jlm_dynamic_vocab): O(L) words per sentence instead of O(L^2).

        -> (seq_words, init_range, delta_words, delta_off, lv_final): init list of cell c =
        seq_words[init_range[2c] : init_range[2c+1]]; delta lists are a CSR over cells;
        lv_final[s][k] is the reference's ``lattice_vocab[k]`` list after the
        decode (original order + appended deltas)."""
        B, F = self.n_sent, self.n_frames
        extra = None
        if samples:
            if random_sampling:      # one draw per sentence, in sentence order, like sentence-at-a-time calls of the reference
                extra = [[int(x) for x in np.random.randint(vocab_len, size=samples)] for _ in range(B)]
            elif top_sampling:
                extra = [list(range(samples))] * B
        lib = host_lib()
        if lib is not None and self.builder.use_native:
            return self._dynamic_vocab_native(lib, extra)
        seq, rng = [], np.zeros(2 * F * B, dtype=np.int32)
        delta = [[] for _ in range(F * B)]
        for s in range(B):
            lv, d = self._dyn_lists_python(s, extra[s] if extra else [])
            L = int(self.sent_len[s])
            base = len(seq)
            lv0 = sorted(lv[0])
            dup = [x for i, x in enumerate(lv0) if i and lv0[i - 1] == x]
            seq += dup + sorted(set(lv0))
            cum = [len(seq)]
            for i in range(1, L + 1):
                delta[i * B + s] = d[i]
                seq += d[i]
                cum.append(len(seq))
            for k in range(L):
                c = k * B + s
                rng[2 * c], rng[2 * c + 1] = (base if k == 0 else base + len(dup)), cum[k + 1]
        dw, do = _csr(delta)
        return np.asarray(seq, dtype=np.int32), rng, dw, do, _LazyFinal(self, extra)

    def dynamic_vocab_compat(self, seg_starts, samples=0, top_sampling=False, random_sampling=False, vocab_len=None):
        """dynamic_vocab() for DynamicDecoder.compat_quirks on a SEGMENTED model (D-softmax / D-softmax*): the lists in
        the reference's own order plus the words whose weight rows the reference actually reads.

        The reference's ``project(state, vocab)`` returns the columns of a vocabulary subset segment by segment
        (model.py:152-158,168-179) while ``DynamicDecoder`` indexes them -- and the model adds the bias -- in list order
        (decoder_dynamic.py:76,130,172).  The list a frame's rows are first normalised over is
        ``L = lattice_vocab[k] + sorted(new words of frame k+1)`` (decoder_dynamic.py:30-46,112-127): position j of it
        gets the weight row of ``P[j]``, P = L stably partitioned by segment, and the bias of ``L[j]``.  Words appended
        at later frames come from ``project(state, sorted(missing))`` -- sorted = segment-major: those are consistent.

        -> (init_words, init_range, delta_words, delta_off, lv_final, init_weight_words, edge_weight_words): init list of
        cell c = init_words[init_range[2c] : init_range[2c+1]] (bias / identity) beside init_weight_words (weight row);
        edge_weight_words is parallel to ``sg_word``.  Host-side Python, sentence by sentence: a compatibility mode."""
        import bisect
        B, F = self.n_sent, self.n_frames
        extra = None
        if samples:
            if random_sampling:
                extra = [[int(x) for x in np.random.randint(vocab_len, size=samples)] for _ in range(B)]
            elif top_sampling:
                extra = [list(range(samples))] * B
        starts = list(seg_starts)
        seg_of = lambda w: bisect.bisect_right(starts, w) - 1
        di, diw = [], []
        rng = np.zeros(2 * F * B, dtype=np.int32)
        delta = [[] for _ in range(F * B)]
        sg_word, sg_off = np.asarray(self.sg_word), np.asarray(self.sg_off)
        sgw = sg_word.astype(np.int32).copy()
        for s in range(B):
            lv, d = self._dyn_lists_python(s, extra[s] if extra else [])
            L = int(self.sent_len[s])
            for i in range(1, L + 1):
                delta[i * B + s] = d[i]
            for k in range(L):
                Lk = [int(w) for w in lv[k]] + d[k + 1]
                by_seg = [[] for _ in starts]
                for w in Lk:
                    by_seg[seg_of(w)].append(w)
                Pk = [w for seg in by_seg for w in seg]
                c = k * B + s
                rng[2 * c] = len(di)
                di += Lk
                diw += Pk
                rng[2 * c + 1] = len(di)
                first = {}
                for j, w in enumerate(Lk):
                    first.setdefault(w, j)                    # list.index: the first occurrence
                for e in range(int(sg_off[c]), int(sg_off[c + 1])):
                    j = first.get(int(sg_word[e]))
                    if j is not None:
                        sgw[e] = Pk[j]
        dw, do = _csr(delta)
        return (np.asarray(di, dtype=np.int32), rng, dw, do, _LazyFinal(self, extra), np.asarray(diw, dtype=np.int32), sgw)

    def dynamic_init_list(self, dyn, k, s):
        """The init list of cell (k, s) out of dynamic_vocab()'s result (tests, debugging)."""
        c = k * self.n_sent + s
        return dyn[0][int(dyn[1][2 * c]):int(dyn[1][2 * c + 1])].tolist()

    def _dyn_lists_python(self, s, extra):
        L = int(self.sent_len[s])
        fw = [sorted(int(self.node_word[n]) for n in self.frame_nodes(s, f)) for f in range(L + 1)]
        lv = {0: list(fw[0]) + list(extra)}
        for i in range(1, L + 1):
            lv[i] = sorted(set(lv[i - 1]) | set(fw[i]))
        d = {i: sorted(set(lv[i]) - set(lv[i - 1])) for i in range(1, L + 1)}
        return lv, d

    def dynamic_final_vocab(self, s, extra):
        """The reference's ``lattice_vocab`` dict of sentence ``s`` after its decode
        (original per-frame lists + the appended deltas, decoder_dynamic.py:112-127)."""
        lv, d = self._dyn_lists_python(s, extra)
        L = int(self.sent_len[s])
        final = {}
        for k in range(L + 1):
            final[k] = list(lv[k])
            for i in range(k + 1, L + 1):
                final[k] += d[i]
        return final

    def _dynamic_vocab_native(self, lib, extra):
        B, F = self.n_sent, self.n_frames
        ncell = F * B
        ex_ids = ex_off = None
        if extra:
            ex_off = np.zeros(B + 1, dtype=np.int32)
            np.cumsum([len(e) for e in extra], out=ex_off[1:])
            ex_ids = np.fromiter((x for e in extra for x in e), dtype=np.int32, count=int(ex_off[-1]))
        rng = np.zeros(2 * ncell, dtype=np.int32)
        do = np.zeros(ncell + 1, dtype=np.int32)
        dtot = np.zeros(1, dtype=np.int64)
        nw = np.ascontiguousarray(self.node_word)
        n_extra = int(ex_off[-1]) if ex_off is not None else 0
        scap, dcap = self.n_nodes + 2 * n_extra + B, self.n_nodes + n_extra + B     # upper bounds: one pass
        while True:
            sw = np.empty(scap, dtype=np.int32)
            dw = np.empty(dcap, dtype=np.int32)
            n = lib.jlm_dynamic_vocab(_ptr(nw), _ptr(self.end_off), _ptr(self.sent_len), B, F,
                                      _ptr(ex_ids) if ex_ids is not None else None,
                                      _ptr(ex_off) if ex_off is not None else None, scap, dcap, _ptr(sw), _ptr(rng),
                                      _ptr(dw), _ptr(do), _ptr(dtot), self.builder.n_threads)
            if n <= scap and int(dtot[0]) <= dcap:
                break
            scap, dcap = max(scap, int(n)), max(dcap, int(dtot[0]))
        return sw[:int(n)], rng, dw[:int(dtot[0])], do, _LazyFinal(self, extra)


class _LazyFinal:
    """lv_final[s] computed on demand (the decoders only expose the last sentence's)."""

    def __init__(self, lat, extra):
        self.lat, self.extra = lat, extra

    def __len__(self):
        return self.lat.n_sent

    def __getitem__(self, s):
        if s < 0:
            s += self.lat.n_sent
        return self.lat.dynamic_final_vocab(s, self.extra[s] if self.extra else [])


class _LazyLists:
    """list-of-lists view over a CSR pair (only materialises what is asked for)."""

    def __init__(self, flat, off):
        self.flat, self.off = flat, off

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if i < 0:
            i += len(self)
        return self.flat[self.off[i]:self.off[i + 1]].tolist()


def _csr(lists):
    off = np.zeros(len(lists) + 1, dtype=np.int32)
    if lists:
        np.cumsum([len(x) for x in lists], out=off[1:])
    flat = np.fromiter((v for l in lists for v in l), dtype=np.int32, count=int(off[-1]))
    return flat, off
