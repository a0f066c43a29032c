"""Wall time of one BatchLattice build (256 x 20-kana sentences, configs[1] lexicon) against the native builder's thread count."""
import os, sys, tempfile, time, pickle
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jlm_amd import config as jconfig, synth
from jlm_amd.lattice import LatticeBuilder, BatchLattice
from jlm_amd.decoder import Decoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, lex, rd, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
with open(os.path.join(root, 'data', 'lexicon.pkl'), 'rb') as f: full_lexicon = pickle.load(f)
with open(os.path.join(root, 'data', 'reading_dict.pkl'), 'rb') as f: full_rd = pickle.load(f)
d = Decoder.__new__(Decoder); d.config = jconfig.load_config_dict(1); d._load_vocab()
b = LatticeBuilder(full_lexicon, full_rd, d.w2i)
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
for nt in (1, 2, 4, 8, 16):
    b.n_threads = nt
    BatchLattice(b, sents, 10)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); lat = BatchLattice(b, sents, 10); ts.append(time.perf_counter() - t)
    ts.sort()
    print("n_threads %2d: BatchLattice median %.3f ms  min %.3f ms  (%d nodes)" % (nt, ts[15] * 1e3, ts[0] * 1e3, lat.n_nodes))
