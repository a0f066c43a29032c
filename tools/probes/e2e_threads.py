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
dec.decode_batch(sents * 6, beam_width=10)
N = 40
res = {}
variants = [tuple(int(v) for v in x.split(",")) for x in sys.argv[1:]] or [(3, 4), (3, 2), (3, 1), (2, 2), (2, 4), (4, 1)]
for rnd in range(4):
    for (w, nt) in variants:
        dec.prefetch_workers = w; dec._pool = None; dec._pool1 = None
        dec._builder.n_threads = nt
        dec.decode_batch(sents * 4, beam_width=10)
        torch.cuda.synchronize(); t = time.perf_counter()
        dec.decode_batch(sents * N, beam_width=10)
        torch.cuda.synchronize()
        res.setdefault((w, nt), []).append((time.perf_counter() - t) / N * 1e3)
for k, v in res.items():
    print("prefetch workers %d, lattice threads %d: median %.3f min %.3f ms/step" % (k[0], k[1], np.median(v), min(v)))
