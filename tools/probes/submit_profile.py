"""where does the host time of one eager submit() go?  (cProfile over 20 submits)"""
import cProfile, io, os, pstats, sys, tempfile, time
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
for _ in range(6): eng.collect(eng.submit(lat, "static", topN=10))
t = time.perf_counter()
for _ in range(20):
    tk = eng.submit(lat, "static", topN=10)
    t1 = time.perf_counter()
    eng.collect(tk)
torch.cuda.synchronize()
pr = cProfile.Profile()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); pr.enable()
    tk = eng.submit(lat, "static", topN=10)
    pr.disable(); ts.append(time.perf_counter() - t0)
    eng.collect(tk)
print("submit: median %.2f ms" % (sorted(ts)[10] * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
