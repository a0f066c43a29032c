"""Incremental-vocabulary-selection decoder on MI355X.

Counterpart of ``DynamicDecoder`` (reference decoder/decoder_dynamic.py:18-194):
same constructor and ``decode`` signature, plus ``decode_batch``.  The
per-frame growing vocabulary becomes an online log-sum-exp extension on the
device (jlm_wordlist_lse with merge=1) and the "re-score every path from the
head" fix-up (decoder_dynamic.py:150-175) becomes a per-sentence scan over the
back-pointer chain inside jlm_beam_step (mode 2).

As in the reference, ``vocab_select=True`` is required (without it the
reference raises TypeError at decoder_dynamic.py:114).  With D-softmax /
D-softmax* models the reference mis-assigns softmax columns in this decoder
(SURVEY.md 8 a16); this implementation computes the self-consistent result
instead and parity is pinned on tied-softmax models.
"""
from collections import deque

from .decoder import Decoder
from .lattice import BatchLattice


class DynamicDecoder(Decoder):
    dynamic = True

    def __init__(self, experiment_id=0, comp=0, device=None):
        super(DynamicDecoder, self).__init__(experiment_id=experiment_id, comp=comp, device=device)
        print('Dynamic RNN decoder loaded')
        self.perf_log_fix_vocab = []
        self.perf_log_fix_lattice_path_prob = []

    def decode_batch(self, inputs, topN=10, beam_width=10, vocab_select=False, samples=0, top_sampling=False,
                     random_sampling=False):
        if beam_width is None:
            raise ValueError("beam_width=None (unpruned search) is not supported by the incremental decoder's GPU path")
        if not 1 <= int(beam_width) <= 64:
            raise ValueError("beam_width must be 1..64 on the GPU path (one wave lane per surviving hypothesis)")
        if not vocab_select:
            raise TypeError("'NoneType' object is not subscriptable")      # decoder_dynamic.py:114
        inputs = list(inputs)
        if not inputs:
            return []
        if any(len(x) == 0 for x in inputs):
            keep = [i for i, x in enumerate(inputs) if len(x)]
            sub = self.decode_batch([inputs[i] for i in keep], topN, beam_width, vocab_select, samples, top_sampling,
                                    random_sampling) if keep else []
            res = [[(0.0, [])] for _ in inputs]
            for i, r in zip(keep, sub):
                res[i] = r
            return res
        out, inflight = [None] * len(inputs), deque()
        chunks = self._chunks(inputs, beam_width, reorder=not (samples and random_sampling))

        def finish(item):
            idx, ticket = item
            for j, r in zip(idx, self._engine.collect(ticket)):
                out[j] = r
            self._log_perf()
            # perf_timing: HIP events around the merge of a frame's new words into the older rows (the reference's
            # "vocab fix", decoder_dynamic.py:112-148) and around the re-scoring beam step ("lattice path fix", :150-175)
            if self.perf_timing and self._engine.last_fix_timing:
                for t_vocab, t_path in self._engine.last_fix_timing:
                    self.perf_log_fix_vocab.append(t_vocab)
                    self.perf_log_fix_lattice_path_prob.append(t_path)

        def prepare(idx):
            lat = BatchLattice(self._builder, [inputs[j] for j in idx], beam_width)
            return (idx, lat) + tuple(lat.dynamic_vocab(samples, top_sampling, random_sampling, len(self.w2i)))

        workers = 1 if (samples and random_sampling) else self.prefetch_workers
        last_lv = None
        for idx, lat, iw, io, dw, do, lv_final in self._prefetched(prepare, chunks, workers):
            self.last_lattice = lat
            if (len(inputs) - 1) in idx:
                last_lv = lv_final[idx.index(len(inputs) - 1)]
            inflight.append((idx, self._engine.submit(lat, "dynamic", dyn_lists=(iw, io, dw, do), topN=topN,
                                                      timing=self.perf_timing)))
            if len(inflight) > self.pipeline_depth:
                finish(inflight.popleft())
        while inflight:
            finish(inflight.popleft())
        self.lattice_vocab = last_lv              # the reference leaves the LAST sentence's dict behind
        self.perf_sen += len(inputs)
        return out
