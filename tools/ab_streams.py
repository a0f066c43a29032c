"""Interleaved in-process A/B of (streams, batches in flight, lse_share_pct): device-resident loop and strings -> strings.
usage: ab_streams.py "streams,depth,share" ...     e.g.  2,2,66 3,3,66 3,3,50"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import jlm_amd
from collections import deque
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
variants = [tuple(int(v) for v in x.split(",")) for x in sys.argv[1:]] or [(2, 2, 66), (3, 3, 66), (3, 3, 50), (2, 3, 66)]
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, os.environ.get("FIXTURE", "mid-vtable"))
jconfig.set_root(root)
dec = Decoder(1)
eng = dec._engine
B = int(os.environ.get("BATCH", "256"))
dec.max_batch = B
sents = synth.make_sentences(B, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)


def setv(v):
    s, d, sh = v
    if eng.n_streams != s:
        torch.cuda.synchronize()
        eng.n_streams, eng._streams, eng._rr = s, [], 0
    dec.pipeline_depth = d
    eng.lse_share_pct = sh


def run(n):
    q = deque()
    for _ in range(n):
        q.append(eng.submit(lat, "static", topN=10))
        if len(q) > dec.pipeline_depth:
            eng.collect(q.popleft())
    while q:
        eng.collect(q.popleft())


dev = {k: [] for k in variants}
e2e = {k: [] for k in variants}
for k in variants:
    setv(k)
    run(8)
    dec.decode_batch(sents * 6, beam_width=10)
N = int(os.environ.get("STEPS", "24"))
for rnd in range(5):
    for k in variants:
        setv(k)
        run(4)
        torch.cuda.synchronize(); t = time.perf_counter()
        run(N)
        torch.cuda.synchronize()
        dev[k].append((time.perf_counter() - t) / N * 1e3)
        torch.cuda.synchronize(); t = time.perf_counter()
        dec.decode_batch(sents * N, beam_width=10)
        torch.cuda.synchronize()
        e2e[k].append((time.perf_counter() - t) / N * 1e3)
for k in variants:
    print("streams %d in-flight %d share %3d%%  device-resident ms/step median %.3f min %.3f | strings->strings median %.3f min %.3f" % (
        k + (np.median(dev[k]), min(dev[k]), np.median(e2e[k]), min(e2e[k]))))
