"""Interleaved in-process A/B of Decoder.collector_thread (finish chunks on their own thread) at 40 and 200 chunks per call."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 8, beam_width=10)
res = {}
for rnd in range(4):
    for N in (40, 200):
        for ct in (True, False):
            dec.collector_thread = ct
            torch.cuda.synchronize(); t = time.perf_counter()
            dec.decode_batch(sents * N, beam_width=10)
            torch.cuda.synchronize()
            res.setdefault((N, ct), []).append((time.perf_counter() - t) / N * 1e3)
for k, v in sorted(res.items()):
    print("chunks %3d  collector thread %-5s  median %.3f  min %.3f ms/step" % (k[0], k[1], np.median(v), min(v)))
