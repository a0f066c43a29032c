"""reference decoder/model_ngram.py -> jlm_amd.model_ngram"""
from jlm_amd.model_ngram import NGramModel  # noqa: F401
