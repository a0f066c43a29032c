"""reference train/data.py (Vocab) -> jlm_amd.data"""
from jlm_amd.data import Vocab  # noqa: F401


class CharVocab(Vocab):
    def __init__(self, *a, **k):
        raise NotImplementedError("CharVocab / char-RNN is outside the scope of this build (SURVEY.md 8f)")
