"""reference train/data.py (Vocab, CharVocab) -> jlm_amd.data"""
from jlm_amd.data import Vocab, CharVocab  # noqa: F401
