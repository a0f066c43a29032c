"""Vocabulary over the frequency-sorted lexicon.

Counterpart of ``Vocab`` in the reference (train/data.py:15-26): index 0 is
``<unk>``, index j>=1 is ``lexicon[j-1]``, and only the first ``size-1``
lexicon entries are in-vocabulary.  ``CharVocab`` (train/data.py:28-47) adds the
character index of a character model on top of it.  Everything else in the
reference's train/data.py is training-only and out of scope.
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


class CharVocab(Vocab):
    """reference train/data.py:28-47: the word vocabulary plus ``c2i`` / ``i2c`` over the characters of the in-vocabulary words'
    display strings -- ``<unk>`` 0, ``<eos>`` 1, then first-occurrence order over ``lexicon[2:]`` (the two entries in front are
    ``<unk>`` and the lexicon's first, ``<eos>``).  ``len()`` is the number of characters: the rows of a character model's softmax."""

    def __init__(self, size, lexicon=None):
        super(CharVocab, self).__init__(size, lexicon)
        self.c2i = {"<unk>": 0, "<eos>": 1}
        for item in self.lexicon[2:]:
            for c in item[0].split("/")[0]:
                if c not in self.c2i:
                    self.c2i[c] = len(self.c2i)
        self.i2c = {v: k for k, v in self.c2i.items()}

    def __len__(self):
        return len(self.c2i)
