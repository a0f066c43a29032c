"""reference decoder/decoder.py -> jlm_amd.decoder (per-frame timing on: the reference's eval.py prints perf_log_*)"""
from jlm_amd.decoder import Decoder as _Decoder, Node  # noqa: F401
from jlm_amd.decoder_char import CharRNNDecoder as _CharRNNDecoder


class Decoder(_Decoder):
    def __init__(self, *a, **k):
        super(Decoder, self).__init__(*a, **k)
        self.perf_timing = True


class CharRNNDecoder(_CharRNNDecoder):
    def __init__(self, *a, **k):
        super(CharRNNDecoder, self).__init__(*a, **k)
        self.perf_timing = True
