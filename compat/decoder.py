"""reference decoder/decoder.py -> jlm_amd.decoder"""
from jlm_amd.decoder import Decoder, CharRNNDecoder, Node  # noqa: F401
