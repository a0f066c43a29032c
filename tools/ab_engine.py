"""A/B of engine options in ONE process, interleaved (the chip's clocks drift between runs and boxes, so
separate bench runs cannot resolve a few per cent): pipelined submit/collect as Decoder.decode_batch does.
usage: ab_engine.py [fixture] [static|static-vs|dynamic]"""
import os, sys, time, tempfile
from collections import deque
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.decoder_dynamic import DynamicDecoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
fixture = sys.argv[1] if len(sys.argv) > 1 else "mid-vtable"
mode = sys.argv[2] if len(sys.argv) > 2 else "static"
cfg, _l, _r, al = synth.build_fixture(root, fixture)
jconfig.set_root(root)
dec = (DynamicDecoder if mode == "dynamic" else Decoder)(1); dec.perf_timing = False
eng = dec._engine; eng.MAX_PLANS = 8
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
from jlm_amd.lattice import BatchLattice
lat = BatchLattice(dec._builder, sents, 10)
kw = {}
if mode == "static-vs":
    words, off, lists = lat.static_vocab(0, False, False, len(dec.w2i)); kw = dict(vocab=(words, off))
if mode == "dynamic":
    kw = dict(dyn_lists=lat.dynamic_vocab(0, False, False, len(dec.w2i))[:4])
kind = "dynamic" if mode == "dynamic" else "static"

def run_pipe(n, depth):
    q = deque()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        q.append(eng.submit(lat, kind, topN=10, **kw))
        if len(q) > depth: eng.collect(q.popleft())
    while q: eng.collect(q.popleft())
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

variants = {"1 stream": (1, False, True, True), "1 stream, no side stream": (1, False, True, False), "2 streams": (2, False, True, True), "2 streams, no side stream": (2, False, True, False),
            "3 streams": (3, False, True, True), "3 streams, no side stream": (3, False, True, False),
            "4 streams, no side stream": (4, False, True, False),
            "2 streams graph": (2, True, True, True), "2 streams, python loop": (2, False, False, True)}
only = os.environ.get("AB_ONLY")
if only:
    variants = {k: v for k, v in variants.items() if any(o.strip() == k for o in only.split(";"))}
def run(v):
    eng.n_streams, eng.use_graph, eng.native_loop, eng.use_side = v
    eng.graph_full = True            # let the variants decide
    if len(eng._streams) < eng.n_streams:
        eng._streams += [torch.cuda.Stream() for _ in range(eng.n_streams - len(eng._streams))]
    eng._rr = 0
    return run_pipe(12, v[0])
for _ in range(4):
    for v in variants.values(): run(v)
res = {k: [] for k in variants}
for rep in range(8):
    for k, v in variants.items(): res[k].append(run(v))
print("streams created:", len(eng._streams), "plans:", len(eng.plans))
for k, v in res.items():
    print("%-30s median %.3f ms/step  min %.3f  (%s)" % (k, np.median(v), min(v), " ".join("%.2f" % x for x in v)))
