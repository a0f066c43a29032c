"""ms per step of Decoder.decode_batch(256-sentence chunks x K) for K = 5 .. 80, repeated: the fixed cost of a call (pipeline fill and drain)."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 12, beam_width=10)
for K in (5, 10, 20, 20, 20, 40, 40, 80, 20, 10, 5):
    torch.cuda.synchronize(); t = time.perf_counter()
    dec.decode_batch(sents * K, beam_width=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("K = %3d: %7.2f ms per call, %.3f ms per step" % (K, dt * 1e3, dt / K * 1e3))
