"""Where the host's time goes in the strings -> strings pipeline (config 2 shape): per-chunk wall time of the main thread's
pieces (waiting for a lattice, submit, collect) and a cProfile of the main thread.  GPU box only."""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder

root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 6, beam_width=10)
N = 40
acc = {"submit": 0.0, "collect": 0.0, "prepare": 0.0}
eng = dec._engine
o_submit, o_collect = eng.submit, eng.collect


def w_submit(*a, **k):
    t = time.perf_counter(); r = o_submit(*a, **k); acc["submit"] += time.perf_counter() - t; return r


def w_collect(*a, **k):
    t = time.perf_counter(); r = o_collect(*a, **k); acc["collect"] += time.perf_counter() - t; return r


eng.submit, eng.collect = w_submit, w_collect
import jlm_amd.decoder as D
o_bl = D.BatchLattice


def w_bl(*a, **k):
    t = time.perf_counter(); r = o_bl(*a, **k); acc["prepare"] += time.perf_counter() - t; return r


D.BatchLattice = w_bl
for rnd in range(3):
    for k in acc: acc[k] = 0.0
    torch.cuda.synchronize(); t = time.perf_counter()
    dec.decode_batch(sents * N, beam_width=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("round %d: %.3f ms/step; main thread submit %.3f collect %.3f ms/step; lattice build (worker threads) %.3f ms/step" % (
        rnd, dt / N * 1e3, acc["submit"] / N * 1e3, acc["collect"] / N * 1e3, acc["prepare"] / N * 1e3))
eng.submit, eng.collect = o_submit, o_collect
D.BatchLattice = o_bl
pr = cProfile.Profile()
pr.enable()
dec.decode_batch(sents * N, beam_width=10)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
