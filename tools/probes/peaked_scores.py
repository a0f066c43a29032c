"""Path-score error of the decode against the oracle as the output embeddings are scaled up (logits x scale: peaked distributions, what a
trained model has and the synthetic fixtures do not), with the normaliser on mixed rows and on split rows (JLM_LSE_MIXED=0).
usage: python tools/probes/peaked_scores.py [scale]"""
import os, sys, tempfile, pickle
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from jlm_amd import config as jconfig, synth
from oracle import jlm_oracle as orc
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
root = os.path.join(tempfile.gettempdir(), "jlm_peaked_%g" % scale)
cfg, _l, _r, al = synth.build_fixture(root, "wide-vtable")
wp = os.path.join(root, "train", "experiments", "1", "weights", "lstm_weights.pkl")
w = pickle.load(open(wp, "rb"))
for key in list(w):
    if key.startswith("LM"):
        w[key] = [b * np.float32(scale) for b in w[key]] if isinstance(w[key], list) else w[key] * np.float32(scale)
pickle.dump(w, open(wp, "wb"))
jconfig.set_root(root)
from jlm_amd.decoder import Decoder
sents = synth.make_ragged_sentences(24, 4, 16, seed=5, alphabet=al)
o = orc.OracleDecoder(root, 1)
want = [o.decode(s, beam_width=8) for s in sents]
for mixed in ("1", "0"):
    os.environ["JLM_LSE_MIXED"] = mixed
    d = Decoder(1)
    got = d.decode_batch(sents, beam_width=8)
    worst, same1, samen = 0.0, 0, 0
    for g, wv in zip(got, want):
        same1 += g[0][1] == wv[0][1]
        samen += [x for _, x in g] == [x for _, x in wv]
        worst = max(worst, max(abs(a[0] - b[0]) for a, b in zip(g, wv)))
    top = max(abs(wv[0][0]) for wv in want)
    print("LM x %g, mixed=%s (idx %s): 1-best %d/%d, n-best %d/%d, max |score diff| %.2e (scores up to %.1f)" % (scale, mixed, d.model.dev.mixed_idx, same1, len(sents), samen, len(sents), worst, top))
