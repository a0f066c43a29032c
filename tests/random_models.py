"""Models nobody tuned a kernel for (shared by tests/test_gpu_random_models.py and the CPU suite, which runs a few draws on the
numpy double of the C ABI).  Seeded random draws of the model shape (vocabulary size, hidden size, embedding width,
projection mode, segment widths and cuts), the weight scale, the beam, the decoder kind and its selected-vocabulary options, each
decoded on ragged sentences and compared with the oracle: identical 1-best, scores within the suite's tolerance, the step logits of
the predict API within 1e-4.  The shapes fall on every branch of the loaders (split rows with and without bias columns, mixed
rows, hybrid launches, padded contractions, the generic kernels) -- whatever a draw lands on must either agree or raise at load
with a message that says why (reference decoder/model.py:106-198, decoder/decoder.py:79-241, decoder/decoder_dynamic.py:30-194)."""
import numpy as np
import pytest

from jlm_amd import config as jconfig, synth       # noqa: E402
from oracle import jlm_oracle as orc               # noqa: E402

SEEDS = list(range(24))


def draw(seed):
    rng = np.random.RandomState(1000 + seed)
    V = int(rng.randint(300, 3200))
    H = int(rng.choice([32, 64, 96, 128, 160, 256]))
    mode = str(rng.choice(["tied", "untied", "dsoftmax", "vtable"]))
    # segment widths: multiples of 4 ... odd ones, cuts anywhere
    n_seg = int(rng.randint(1, 5))
    widths = sorted((int(rng.choice([8, 12, 20, 36, 50, 64, 100, 128, 200, 256])) for _ in range(n_seg)), reverse=True)
    cuts = sorted(int(x) for x in rng.choice(np.arange(40, V - 40), size=n_seg - 1, replace=False)) if n_seg > 1 else []
    bounds = [0] + cuts + [None]
    segs = [(widths[i], bounds[i], bounds[i + 1]) for i in range(n_seg)]
    E = int(rng.choice([16, 20, 32, 50, 64, 100, 128, 256]))
    if mode == "vtable":
        E = widths[0]
    # (the incremental decoder takes tied models; an untied projection has no vocabulary subset: decoder/model.py:166-170)
    kinds = {"tied": ["static", "static-vs", "dynamic", "dynamic"], "untied": ["static"]}.get(mode, ["static", "static-vs"])
    kind = str(rng.choice(kinds))
    kw = {}
    if kind != "static":
        kw["vocab_select"] = True
        pick = int(rng.randint(0, 3))
        if pick == 1:
            kw.update(samples=int(rng.randint(5, 60)), top_sampling=True)
    beam = int(rng.choice([1, 3, 5, 10, 17, 40]))
    scale = float(rng.choice([0.1, 0.2, 0.35])) * (64.0 / H) ** 0.5
    alphabet = int(rng.choice([8, 12, 20]))
    return dict(V=V, H=H, E=E, mode=mode, segs=segs, kind=kind, kw=kw, beam=beam, scale=scale, alphabet=alphabet)


def check(seed, root):
    p = draw(seed)
    cfg = synth.make_config(p["V"], p["H"], p["E"], p["mode"], p["segs"])
    synth.write_lexicon(root, p["V"], alphabet=p["alphabet"])
    synth.write_experiment(root, 1, cfg, scale=p["scale"], seed=20 + seed)
    jconfig.set_root(root)
    from jlm_amd.decoder import Decoder
    from jlm_amd.decoder_dynamic import DynamicDecoder
    dyn = p["kind"] == "dynamic"
    d = (DynamicDecoder if dyn else Decoder)(1)
    o = (orc.OracleDynamicDecoder if dyn else orc.OracleDecoder)(root, 1)
    sents = synth.make_ragged_sentences(6, 1, 16, seed=seed, alphabet=p["alphabet"])
    got = d.decode_batch(sents, beam_width=p["beam"], **p["kw"])
    for s, g in zip(sents, got):
        w = o.decode(s, beam_width=p["beam"], **p["kw"])
        assert len(g) == len(w), (p, s)
        assert g[0][1] == w[0][1], (p, s, g[0], w[0])
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in w], rtol=2e-6, atol=2e-5, err_msg=str((p, s)))
    # the predict API on an odd row count
    rng = np.random.RandomState(seed)
    R = int(rng.randint(1, 40))
    idx = [int(x) for x in rng.randint(0, p["V"], size=R)]
    h0, c0 = rng.normal(0, 0.3, (R, p["H"])), rng.normal(0, 0.3, (R, p["H"]))
    (pred, y, _a, _b), h, c = d.model.predict_with_context(idx, h0, c0)
    pr, yr, hr, cr, _, _ = o.model.predict(idx, h0, c0)
    assert np.abs(y - yr).max() <= 1e-4 * max(np.abs(yr).max(), 1.0), p
    np.testing.assert_allclose(h, hr, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c, cr, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pred, pr, rtol=2e-4, atol=1e-7)
