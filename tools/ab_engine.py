"""A/B of engine options in ONE process, interleaved (the chip's clocks drift between runs, so
separate bench runs cannot resolve a few per cent): configs[1] batch, pipelined submit/collect."""
import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, sys.argv[1] if len(sys.argv) > 1 else "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.perf_timing = False
eng = dec._engine
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)

def run_pipe(n=12):
    prev = None
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        tk = eng.submit(lat, "static", topN=10)
        if prev is not None: eng.collect(prev)
        prev = tk
    eng.collect(prev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run_pipe2(n=12, depth=2):
    """consecutive steps on alternating streams: the device work of step i+1 may overlap step i's"""
    from collections import deque
    q = deque()
    torch.cuda.synchronize(); t = time.perf_counter()
    for it in range(n):
        with torch.cuda.stream(streams[it % 2]):
            q.append(eng.submit(lat, "static", topN=10))
        if len(q) > depth: eng.collect(q.popleft())
    while q: eng.collect(q.popleft())
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

variants = {"eager+side": (False, True), "2 streams d2": (False, True, 2), "2 streams d3": (False, True, 3), "eager": (False, False)}
def run(v):
    eng.use_graph, eng.use_side = v[0], v[1]
    return run_pipe2(12, v[2]) if len(v) > 2 else run_pipe(12)
for _ in range(3):
    for v in variants.values():
        run(v)
res = {k: [] for k in variants}
for rep in range(8):
    for k, v in variants.items():
        res[k].append(run(v))
for k, v in res.items():
    print("%-12s median %.3f ms/step  min %.3f  (%s)" % (k, np.median(v), min(v), " ".join("%.2f" % x for x in v)))
