"""Interleaved in-process A/B of the vocabulary kernel's range count (JLM_LSE_NP): how many CUs it leaves to the other batch
in flight.  usage: ab_np.py [np ...]   (0 = the launcher's own choice)"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import jlm_amd
from collections import deque
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
nps = [int(x) for x in sys.argv[1:]] or [0, 22, 20, 19, 18, 17, 16, 14]
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, os.environ.get("FIXTURE", "mid-vtable"))
jconfig.set_root(root)
dec = Decoder(1)
eng = dec._engine
B = int(os.environ.get("BATCH", "256"))
sents = synth.make_sentences(B, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)
os.environ["JLM_LSE_NP8"] = "0"


def run(n):
    q = deque()
    for _ in range(n):
        q.append(eng.submit(lat, "static", topN=10))
        if len(q) > dec.pipeline_depth:
            eng.collect(q.popleft())
    while q:
        eng.collect(q.popleft())


res = {k: [] for k in nps}
for k in nps:
    os.environ["JLM_LSE_NP"] = str(k)
    run(6)
for rnd in range(5):
    for k in nps:
        os.environ["JLM_LSE_NP"] = str(k)
        run(3)
        torch.cuda.synchronize(); t = time.perf_counter()
        run(24)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t) / 24 * 1e3)
for k in nps:
    print("NP=%2d  ms/step median %.3f  min %.3f  (%s)" % (k, np.median(res[k]), min(res[k]), " ".join("%.3f" % x for x in res[k])))
