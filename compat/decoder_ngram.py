"""reference decoder/decoder_ngram.py -> jlm_amd.decoder_ngram"""
from jlm_amd.decoder_ngram import NGramDecoder  # noqa: F401
