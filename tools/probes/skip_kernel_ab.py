"""upper bound of what a kernel costs the pipelined step: replace its C entry point by a no-op
(results are then wrong; timing only).  usage: skip_kernel_ab.py jlm_edge_logits [jlm_beam_step ...]"""
import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import config as jconfig, synth, _lib
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.perf_timing = False
eng = dec._engine
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)
L = _lib.lib()
class Proxy:
    def __init__(self, lib, skip): self._lib, self._skip = lib, set(skip)
    def __getattr__(self, n):
        if n in self._skip: return lambda *a: 0
        return getattr(self._lib, n)
def run_pipe(n=12):
    prev = None
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        tk = eng.submit(lat, "static", topN=10)
        if prev is not None: eng.collect(prev)
        prev = tk
    eng.collect(prev); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
real = _lib.lib
variants = [("all kernels", [])] + [("without " + k, [k]) for k in sys.argv[1:]]
res = {k: [] for k, _ in variants}
for rep in range(6):
    for name, skip in variants:
        _lib.lib = (lambda s=skip: Proxy(L, s)) if skip else real
        run_pipe(6)
        res[name].append(run_pipe(12))
_lib.lib = real
for k, v in res.items():
    print("%-40s median %.3f ms/step  (%s)" % (k, np.median(v), " ".join("%.2f" % x for x in v)))
