"""What a short decode_batch call pays after a warm-up: cProfile of the first 6-chunk call."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.perf_timing = False; dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 4, beam_width=10)
torch.cuda.synchronize()
for rep in range(3):
    pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
    dec.decode_batch(sents * 6, beam_width=10)
    pr.disable(); dt = time.perf_counter() - t
    print("call %d: %.1f ms for 6 chunks; plans %d" % (rep, dt * 1e3, len(dec._engine.plans)))
    if rep == 0:
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
