"""reference decoder/decoder_dynamic.py -> jlm_amd.decoder_dynamic (per-frame timing on: the reference's eval.py prints perf_log_*)"""
from jlm_amd.decoder_dynamic import DynamicDecoder as _DynamicDecoder


class DynamicDecoder(_DynamicDecoder):
    def __init__(self, *a, **k):
        super(DynamicDecoder, self).__init__(*a, **k)
        self.perf_timing = True
