"""Where the calling thread's time goes per chunk, for any decoder kind (static | static-vs | dynamic) on the tied V=50k model:
prepare (worker threads), submit pieces (staging copies, the frame-loop op), collect (event wait, read-out), and the thread's CPU.
usage: python tools/probes/host_profile2.py [static|static-vs|dynamic] [fixture]"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth, ops
from jlm_amd.decoder import Decoder
from jlm_amd.decoder_dynamic import DynamicDecoder
from jlm_amd.engine import DecodeEngine

kind = sys.argv[1] if len(sys.argv) > 1 else "static-vs"
fixture = sys.argv[2] if len(sys.argv) > 2 else "mid-tied"
root = os.path.join(tempfile.gettempdir(), "jlm_dbg_" + fixture)
cfg, _l, _r, al = synth.build_fixture(root, fixture)
jconfig.set_root(root)
dec = (DynamicDecoder if kind == "dynamic" else Decoder)(1)
dec.max_batch = 256
kw = dict(vocab_select=True) if kind != "static" else {}
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 8, beam_width=10, **kw)
N = 40
acc = {}


def wrap(obj, name, key, static=False):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
    setattr(obj, name, staticmethod(g) if static else g)


eng = dec._engine
wrap(eng, "submit", "submit")
wrap(eng, "collect", "collect")
wrap(eng, "_enqueue", "submit/_enqueue")
wrap(DecodeEngine, "_read_out", "collect/read_out", static=True)
be = ops.backend()


class Ops:
    def __getattr__(self, n):
        return getattr(be, n)

    def decode_frames(self, *a):
        t = time.perf_counter()
        r = be.decode_frames(*a)
        acc["submit/decode_frames op"] = acc.get("submit/decode_frames op", 0.0) + time.perf_counter() - t
        return r


    def decode_batch(self, *a):
        t = time.perf_counter()
        r = be.decode_batch(*a)
        acc["submit/decode_batch op"] = acc.get("submit/decode_batch op", 0.0) + time.perf_counter() - t
        return r


ops._backend = Ops()
import jlm_amd.lattice as LT
for nm in ("static_vocab", "dynamic_vocab"):
    wrap(LT.BatchLattice, nm, "prepare/" + nm)
o_init = LT.BatchLattice.__init__


def w_init(self, *a, **k):
    t = time.perf_counter(); o_init(self, *a, **k); acc["prepare/lattice"] = acc.get("prepare/lattice", 0.0) + time.perf_counter() - t


LT.BatchLattice.__init__ = w_init
import jlm_amd.decoder as D, jlm_amd.decoder_dynamic as DD
for rnd in range(3):
    acc.clear()
    torch.cuda.synchronize(); c0 = time.thread_time(); p0 = time.process_time(); t = time.perf_counter()
    dec.decode_batch(sents * N, beam_width=10, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%s round %d: %.3f ms/step; calling thread CPU %.3f ms/step, process CPU %.3f ms/step" % (
        kind, rnd, dt / N * 1e3, (time.thread_time() - c0) / N * 1e3, (time.process_time() - p0) / N * 1e3))
    print("   " + "  ".join("%s %.3f" % (k, v / N * 1e3) for k, v in sorted(acc.items())))
