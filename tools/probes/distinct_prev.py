"""How many DISTINCT predecessor rows do the live rows of a frame have?  (The state half of the LSTM step, h[bp].W_h, is
the same for every hypothesis that extends the same predecessor; only the word differs.)"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
fixture = sys.argv[1] if len(sys.argv) > 1 else "mid-vtable"
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, fixture)
jconfig.set_root(root)
dec = Decoder(1); dec.perf_timing = False
eng = dec._engine
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)
eng.decode(lat, "static", topN=10)
p = eng.last_state
torch.cuda.synchronize()
F, rmax = lat.n_frames, p.rmax
bp = p.bp.cpu().numpy(); cnt = p.cnt.cpu().numpy().reshape(-1, lat.n_sent)
tot_rows = tot_dist = 0
for f in range(1, F - 1):
    rows, dist = 0, 0
    for s in range(lat.n_sent):
        k = int(cnt[f, s])
        g0 = f * rmax + s * lat.beam
        b = bp[g0:g0 + k]
        rows += k; dist += len(np.unique(b))
    tot_rows += rows; tot_dist += dist
    if f in (1, 2, 3, 5, 10, 15, 19): print("frame %2d: %4d live rows, %4d distinct predecessors (%.2f)" % (f, rows, dist, dist / max(rows, 1)))
print("all frames: %d rows, %d distinct predecessors: %.3f" % (tot_rows, tot_dist, tot_dist / tot_rows))
