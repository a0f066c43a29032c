"""reference decoder/decoder_dynamic.py -> jlm_amd.decoder_dynamic"""
from jlm_amd.decoder_dynamic import DynamicDecoder  # noqa: F401
