"""Back-off n-gram model of the reference's baseline decoder (CPU).

Counterpart of ``NGramModel`` (reference decoder/model_ngram.py:11-69): same constructor, ``predict`` and
``evaluate``.  The model file is the SRILM text format with tab-separated fields; a line with fewer than two
fields is not an n-gram.  Costs are natural-log costs, -ln(10^log10p), computed with the reference's own
expression so that path scores agree to the last bit.  This baseline has nothing to put on a GPU: it is a
dictionary look-up per candidate, kept for the evaluation harness' ``-ng`` switch.
"""
import math
import os

from . import config as _config

_SENTENCE_MARKS = ("<s>", "</s>")       # both mean <eos> in the decoders' vocabulary (model_ngram.py:21-28)
UNSEEN_COST = 100.0                      # a word without a unigram (model_ngram.py:62-63)


class NGramModel():
    def __init__(self, ngram_file='lm3', ngram_order=3):
        self.ngram_order = ngram_order
        self.model = self.parse_srilm(os.path.join(_config.data_path, ngram_file))

    @staticmethod
    def _words(ngram):
        return tuple('<eos>' if w in _SENTENCE_MARKS else w for w in ngram.split(' '))

    def parse_srilm(self, file):
        """-> {tuple of words: (cost, back-off cost or None)}; later duplicates replace earlier ones."""
        print('{} loaded'.format(file))
        table = {}
        with open(file, 'r', encoding='utf-8') as f:
            for line in f:
                fields = line.rstrip('\n').split('\t', 2)
                if len(fields) < 2:
                    continue
                backoff = -math.log(10 ** float(fields[2])) if len(fields) > 2 else None
                table[self._words(fields[1])] = (-math.log(10 ** float(fields[0])), backoff)
        print('{} ngrams loaded'.format(len(table)))
        return table

    def predict(self, words, debug=False):
        """Cost of the last word given its history: the longest suffix of the last ``ngram_order`` words that the
        table holds decides (no back-off weights are applied, as in the reference); nothing found: 100."""
        ctx = tuple(words[-self.ngram_order:]) if isinstance(words, list) else tuple(words)
        table = self.model
        for start in range(len(ctx)):
            hit = table.get(ctx[start:])
            if hit is not None:
                if debug:
                    print(ctx[start:])
                return hit[0]
        return UNSEEN_COST

    def evaluate(self, words, debug=False):
        """Total cost of a word sequence that starts a sentence."""
        seq = ['<eos>'] + list(words)
        costs = []
        for end in range(2, len(seq) + 1):
            costs.append(self.predict(seq[:end], debug))
            if debug:
                print(costs[-1])
        return sum(costs)
