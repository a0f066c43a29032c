"""Round 5 race hunt: the same chunk decoded twice on the same plan -- which rows' log-normalisers differ (frame, sentence, slot)?
JLM_MX_WIDE=1 python tools/probes/wide_determinism.py"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
root = os.path.join(tempfile.gettempdir(), "jlm_det")
cfg, _l, _r, al = synth.build_fixture(root, sys.argv[1] if len(sys.argv) > 1 else "wide-vtable")
jconfig.set_root(root)
dec = Decoder(1); eng = dec._engine
sents = synth.make_ragged_sentences(24 * 48, 1, 22, seed=123, alphabet=al)
dec.max_batch = 48
chunks = dec._chunks(sents, 8)
for ci in (0, 5):
    idx = chunks[ci]
    snaps = []
    for rep in range(4):
        dec.decode_batch([sents[j] for j in idx], beam_width=8)
        torch.cuda.synchronize()
        p = eng.last_state
        snaps.append((p.lse.cpu().numpy().copy(), p.part.cpu().numpy().copy() if p.part is not None else None, id(p), p.rmax, p.F))
    f0 = snaps[0][0][0:snaps[0][3]:8][:len(idx)]             # frame 0: one live row per sentence, all from the same state
    vals, cnt = np.unique(f0, return_counts=True)
    print("chunk %d: frame 0 holds %d distinct log-normalisers over %d identical rows: %s" % (ci, len(vals), len(f0), list(zip(["%.10f" % v for v in vals], cnt.tolist()))[:6]))
    print("   rows with the minority value per decode:", [sorted(np.nonzero(sn[0][0:sn[3]:8][:len(idx)] != np.median(sn[0][0:sn[3]:8][:len(idx)]))[0].tolist()) for sn in snaps])
    for rep in range(1, 4):
        a, b = snaps[0][0], snaps[rep][0]
        rmax = snaps[0][3]
        d = np.argwhere((a != b) & np.isfinite(a) & np.isfinite(b))[:, 0]
        print("chunk %d rep %d: same plan %s; %d lse rows differ; (frame, row-in-frame) of the first few: %s; max |diff| %.3g" % (
            ci, rep, snaps[0][2] == snaps[rep][2], len(d), [(int(g // rmax), int(g % rmax)) for g in d[:8]],
            np.abs(a[d] - b[d]).max() if len(d) else 0.0))
