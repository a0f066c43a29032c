import os, sys, time, tempfile
from collections import deque
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.perf_timing = False
eng = dec._engine
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)
def run_pipe(n, depth):
    q = deque()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        q.append(eng.submit(lat, "static", topN=10))
        if len(q) > depth: eng.collect(q.popleft())
    while q: eng.collect(q.popleft())
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("n_streams", eng.n_streams, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
if os.environ.get("PROBE_NO_SIDE"):
    eng.use_side = False
if len(sys.argv) > 1 and sys.argv[1] == "default-first":
    eng.n_streams = 1
    for rep in range(3):
        print("1 stream FIRST: %.3f" % run_pipe(12, 1))
    eng.n_streams = 2
for rep in range(3):
    print("default 2 streams: %.3f" % run_pipe(12, 2), "plans", len(eng.plans), "streams", len(eng._streams))
eng.n_streams = 1
for rep in range(3):
    print("1 stream: %.3f" % run_pipe(12, 1))
eng.n_streams = 2
for rep in range(3):
    print("back to 2 streams: %.3f" % run_pipe(12, 2), "plans", len(eng.plans))
