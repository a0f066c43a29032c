import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
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
def run(tag, n=24):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.decode(lat, "static", topN=10)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(tag, " ".join("%.1f" % x for x in ts))
def run_pipe(tag, n=24):
    ts = []; prev = None
    torch.cuda.synchronize()
    for _ in range(n):
        t = time.perf_counter()
        tk = eng.submit(lat, "static", topN=10)
        if prev is not None: eng.collect(prev)
        prev = tk
        ts.append((time.perf_counter() - t) * 1e3)
    eng.collect(prev)
    print(tag, " ".join("%.1f" % x for x in ts))
run("sync ")
run_pipe("pipe ")
run_pipe("pipe2")
run("sync2")
