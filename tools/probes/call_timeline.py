"""Timeline of ONE decode_batch call of K chunks (default 20): when the first lattice is ready, when each chunk is submitted and
finished -- where a short call's fixed cost goes."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
import jlm_amd.decoder as D
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 5, beam_width=10); dec.decode_batch(sents * 7, beam_width=10)
eng = dec._engine
ev = []
o_submit, o_collect, o_bl = eng.submit, eng.collect, D.BatchLattice
def w_submit(*a, **k):
    t = time.perf_counter(); r = o_submit(*a, **k); ev.append(("submit", t, time.perf_counter())); return r
def w_collect(*a, **k):
    t = time.perf_counter(); r = o_collect(*a, **k); ev.append(("collect", t, time.perf_counter())); return r
def w_bl(*a, **k):
    t = time.perf_counter(); r = o_bl(*a, **k); ev.append(("lattice", t, time.perf_counter())); return r
eng.submit, eng.collect, D.BatchLattice = w_submit, w_collect, w_bl
for rep in range(3):
    ev.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.decode_batch(sents * K, beam_width=10)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("call of %d chunks: %.2f ms = %.3f ms per chunk" % (K, (t1 - t0) * 1e3, (t1 - t0) / K * 1e3))
    if rep == 0:
        cs = [b - t0 for n, a, b in ev if n == "collect"]
        print("  first call, collect completions, ms:", " ".join("%.1f" % (c * 1e3) for c in cs))
        print("  plans:", len(eng.plans))
    if rep == 2:
        for name in ("lattice", "submit", "collect"):
            xs = [(a - t0, b - t0) for n, a, b in ev if n == name]
            print("  %-8s first %.2f..%.2f ms, second %.2f..%.2f, ... last %.2f..%.2f; mean duration %.3f ms" % (
                name, xs[0][0] * 1e3, xs[0][1] * 1e3, xs[1][0] * 1e3, xs[1][1] * 1e3, xs[-1][0] * 1e3, xs[-1][1] * 1e3,
                np.mean([b - a for a, b in xs]) * 1e3))
        cs = [b - t0 for n, a, b in ev if n == "collect"]
        print("  collect completions, ms:", " ".join("%.1f" % (c * 1e3) for c in cs))
