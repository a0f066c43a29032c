"""The n-gram baseline decoder of the evaluation harness (CPU).

Counterpart of ``NGramDecoder`` (reference decoder/decoder_ngram.py:37-125): the same lattice beam search as the
neural decoders with the n-gram cost of a word given the path's last words in place of the LSTM.  Differences of
the reference's own lattice that are kept: no raw-kana ``<unk>`` fallback (a frame no word ends at stays empty,
decoder_ngram.py:58-82), so an input the lexicon cannot cover yields an empty list.

Paths are (cost, history tuple, back pointer) records rather than copied node lists; candidate order (node order,
then the predecessor frame's beam order) and Python's stable sort give the reference's tie-break.
"""
import json
import os
import pickle
import time

from . import config as _config
from .data import Vocab
from .lattice import BatchLattice, LatticeBuilder
from .model_ngram import NGramModel


class NGramDecoder():
    def __init__(self, experiment_id=0, ngram_order=3):
        with open(os.path.join(_config.experiment_path, str(experiment_id), 'config.json'), 'rt') as f:
            self.config = json.loads(f.read())
        vocab = Vocab(self.config['vocab_size'])
        self.i2w, self.w2i = vocab.i2w, vocab.w2i
        with open(os.path.join(_config.root_path, 'data', 'lexicon.pkl'), 'rb') as f:
            self.full_lexicon = pickle.load(f)
        with open(os.path.join(_config.root_path, 'data', 'reading_dict.pkl'), 'rb') as f:
            self.full_reading_dict = pickle.load(f)
        self.model = NGramModel(ngram_file='lm3', ngram_order=ngram_order)
        self._builder = LatticeBuilder(self.full_lexicon, self.full_reading_dict, self.w2i)
        self.perf_sen = 0
        self.perf_log = []

    def _check_oov(self, word):
        return word not in self.w2i

    def _frames(self, text):
        """Per end frame the (start frame, word) pairs of the lattice, in the reference's order; the neural
        decoders' <unk> fallback nodes are not part of this decoder's lattice."""
        lat = BatchLattice(self._builder, [text], 1)
        frames = []
        for f in range(len(text) + 1):
            frames.append([(int(lat.node_start[n]), lat.word_str(n)) for n in lat.frame_nodes(0, f) if lat.node_lex[n] != -2])
        return frames

    def decode(self, input, topN=10, beam_width=10, use_oov=False, vocab_select=False, samples=0, top_sampling=False,
               random_sampling=False):
        order = self.model.ngram_order
        frames = self._frames(input)
        # beam[f] = [(cost, last words of the path, back pointer (frame, slot) or None, word)]
        beam = [[(0.0, ('<eos>',), None, '<eos>')]]
        for f in range(1, len(input) + 1):
            cands = []
            for start, word in frames[f]:
                for slot, (cost, hist, _bp, _w) in enumerate(beam[start]):
                    t0 = time.time()
                    ctx = hist + (word,)
                    step = self.model.predict(ctx[-order:])
                    self.perf_log.append(time.time() - t0)
                    cands.append((cost + step, ctx[-order:], (start, slot), word))
            if beam_width is not None:
                cands.sort(key=lambda c: c[0])
                cands = cands[:beam_width]
            beam.append(cands)
        out = []
        for cost, _hist, bp, word in beam[len(input)][:topN]:
            words = [word]
            while bp is not None:
                _c, _h, bp, w = beam[bp[0]][bp[1]]
                words.append(w)
            out.append((cost, [w for w in reversed(words) if w != '<eos>']))
        self.perf_sen += 1
        return out

    def decode_batch(self, inputs, **kw):
        return [self.decode(x, **kw) for x in inputs]
