"""Vocabulary over the frequency-sorted lexicon.

Counterpart of ``Vocab`` in the reference (train/data.py:15-26): index 0 is
``<unk>``, index j>=1 is ``lexicon[j-1]``, and only the first ``size-1``
lexicon entries are in-vocabulary.  Everything else in the reference's
train/data.py is training-only and out of scope.
"""
import os
import pickle

from . import config as _config


class Vocab(object):
    def __init__(self, size, lexicon=None):
        if lexicon is None:
            with open(os.path.join(_config.data_path, "lexicon.pkl"), "rb") as f:
                lexicon = pickle.load(f)
        self.lexicon = [("<unk>", 0)] + list(lexicon[: size - 1])
        self.w2i = {x[0]: i for i, x in enumerate(self.lexicon)}
        self.i2w = {v: k for k, v in self.w2i.items()}

    def __len__(self):
        return len(self.w2i)
