"""Where the host time of strings -> n-best strings goes: Decoder.decode_batch over N chunks of 256 sentences,
wall clock per chunk and a cProfile of the calling thread.  usage: e2e_profile.py [fixture] [static|static-vs|dynamic] [chunks]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.decoder_dynamic import DynamicDecoder
fixture = sys.argv[1] if len(sys.argv) > 1 else "mid-vtable"
mode = sys.argv[2] if len(sys.argv) > 2 else "static"
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 40
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, fixture)
jconfig.set_root(root)
dec = (DynamicDecoder if mode == "dynamic" else Decoder)(1)
dec.perf_timing = False
dec.max_batch = 256
if os.environ.get('E2E_DEPTH'):
    dec.pipeline_depth = int(os.environ['E2E_DEPTH']); dec._engine.MAX_PLANS = 8
kw = dict(vocab_select=True) if mode != "static" else {}
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
if os.environ.get('E2E_RAGGED'):
    sents = synth.make_ragged_sentences(256, 1, 30, seed=99, alphabet=al)
dec.decode_batch(sents * 4, beam_width=10, **kw)
for n in (6, chunks, chunks, chunks):
    torch.cuda.synchronize(); t = time.perf_counter()
    dec.decode_batch(sents * n, beam_width=10, **kw)
    dt = time.perf_counter() - t
    print("%s %s: %d chunks  %.3f ms/chunk  %.0f chars/s" % (fixture, mode, n, dt / n * 1e3, sum(len(x) for x in sents) * n / dt))
pr = cProfile.Profile()
pr.enable()
dec.decode_batch(sents * chunks, beam_width=10, **kw)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
print("---- by own time")
st.sort_stats("tottime").print_stats(22)
