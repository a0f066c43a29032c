"""Interleaved in-process A/B of DecodeEngine.lse_share_pct (CUs the vocabulary kernel takes while another batch is in flight):
the device-resident loop and the strings -> strings path.  usage: ab_share.py [pct ...]   (0 = all CUs)"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import jlm_amd
from collections import deque
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
shares = [int(x) for x in sys.argv[1:]] or [0, 75, 66, 60, 50]
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, os.environ.get("FIXTURE", "mid-vtable"))
jconfig.set_root(root)
dec = Decoder(1)
eng = dec._engine
B = int(os.environ.get("BATCH", "256"))
dec.max_batch = B
sents = synth.make_sentences(B, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)


def run(n):
    q = deque()
    for _ in range(n):
        q.append(eng.submit(lat, "static", topN=10))
        if len(q) > dec.pipeline_depth:
            eng.collect(q.popleft())
    while q:
        eng.collect(q.popleft())


dev = {k: [] for k in shares}
e2e = {k: [] for k in shares}
for k in shares:
    eng.lse_share_pct = k
    run(6)
    dec.decode_batch(sents * 4, beam_width=10)
N = int(os.environ.get("STEPS", "24"))
for rnd in range(5):
    for k in shares:
        eng.lse_share_pct = k
        run(3)
        torch.cuda.synchronize(); t = time.perf_counter()
        run(N)
        torch.cuda.synchronize()
        dev[k].append((time.perf_counter() - t) / N * 1e3)
        torch.cuda.synchronize(); t = time.perf_counter()
        dec.decode_batch(sents * N, beam_width=10)
        torch.cuda.synchronize()
        e2e[k].append((time.perf_counter() - t) / N * 1e3)
for k in shares:
    print("share %3d%%  device-resident ms/step median %.3f min %.3f | strings->strings median %.3f min %.3f" % (
        k, np.median(dev[k]), min(dev[k]), np.median(e2e[k]), min(e2e[k])))
