"""where do the occasional +60 ms decode calls come from?  (host gc vs device)"""
import gc, os, sys, time, tempfile
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
def steps(n):
    out = []; prev = None
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        tk = eng.submit(lat, "static", topN=10); t1 = time.perf_counter()
        r = eng.collect(tk); t2 = time.perf_counter()
        out.append(((t1 - t) * 1e3, (t2 - t1) * 1e3))
    return out
steps(20)
for mode in ("gc on", "gc off", "gc on"):
    if mode == "gc off": gc.collect(); gc.disable()
    else: gc.enable()
    o = steps(150)
    tot = [a + b for a, b in o]
    bad = [(i, "%.1f+%.1f" % o[i]) for i in range(len(o)) if tot[i] > 1.5 * np.median(tot)]
    print(mode, "median %.2f ms; outliers:" % np.median(tot), bad[:12], "gc counts", gc.get_count())
