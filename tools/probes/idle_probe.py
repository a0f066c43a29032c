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
dec.decode_batch(sents * 5, beam_width=10); dec.decode_batch(sents * 7, beam_width=10)
def call(tag, K=40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dec.decode_batch(sents * K, beam_width=10); torch.cuda.synchronize()
    print("%-34s %.3f ms per chunk" % (tag, (time.perf_counter() - t0) / K * 1e3))
call("first 40-chunk call"); call("second"); call("third")
time.sleep(1.0); call("after 1 s idle"); call("again")
time.sleep(0.2); call("after 0.2 s idle"); call("again")
time.sleep(0.05); call("after 50 ms idle")
