"""ms per step of successive Decoder.decode_batch calls in a fresh process: how many untimed steps a process needs before it is in
its steady state (bench.py's settle calls).  usage: python tools/probes/warmup_curve.py K1,K2,...   (chunks of 256 sentences per call)"""
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
seq = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,20,20,20,20,20").split(",")]
out = []
for K in seq:
    torch.cuda.synchronize(); t = time.perf_counter()
    dec.decode_batch(sents * K, beam_width=10)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t) / K * 1e3)
print(" ".join("%d:%.3f" % (k, x) for k, x in zip(seq, out)))
