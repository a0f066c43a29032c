"""Latency of the reference's own calling pattern: Decoder.decode(one sentence), sentence after sentence
(eval.py:83), with and without the per-frame timing events eval.py's perf logs need."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.decoder_dynamic import DynamicDecoder
fixture = sys.argv[1] if len(sys.argv) > 1 else "mid-tied"
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, fixture)
jconfig.set_root(root)
sents = synth.make_sentences(64, 20, seed=4242, alphabet=al)
if os.environ.get('SS_RAGGED'):
    sents = synth.make_ragged_sentences(200, 3, 45, seed=7, alphabet=al)      # every length bucket, every list-size class
for cls, kw in ((Decoder, {}), (Decoder, dict(vocab_select=True)), (DynamicDecoder, dict(vocab_select=True))):
    dec = cls(1)
    for timing in (True, False):
        dec.perf_timing = timing
        for s in sents[:8]: dec.decode(s, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        for s in sents: dec.decode(s, **kw)
        dt = (time.perf_counter() - t) / len(sents)
        print("%-15s %-22s perf_timing=%-5s  %.2f ms per sentence  %.0f chars/s" % (cls.__name__, kw, timing, dt * 1e3, sum(len(x) for x in sents) / len(sents) / dt))
