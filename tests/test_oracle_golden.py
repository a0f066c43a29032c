"""Pins the CPU oracle (oracle/jlm_oracle.py) to outputs captured from the
reference itself (tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import jlm_oracle as orc
from tests import golden_cases as gc

SHAPED_FULL = [c for c in gc.DECODE_CASES if c[1] in gc.SHAPED_FIXTURES and c[0].endswith("/static")]
FAST_DECODE = [c for c in gc.DECODE_CASES if not c[0].startswith(("mid-tied/static", "big-")) and c not in SHAPED_FULL] + \
              [c for c in gc.DECODE_CASES if c[0] == "mid-tied/static-vs"]
FAST_DECODE = list({c[0]: c for c in FAST_DECODE}.values())


@pytest.mark.parametrize("name", gc.LM_FIXTURES)
def test_lm_steps_match_reference(name, fx, golden_lm):
    f = fx(name)
    dec = orc.OracleDecoder(f["root"], 1)
    lm = dec.model
    for rows in gc.LM_ROWS:
        idx, subset, cols, h0, c0 = gc.lm_inputs(f["cfg"], rows)
        for kind in ("full", "subset"):
            if kind == "subset" and not f["cfg"]["share_embedding"]:
                continue
            vocab = subset if kind == "subset" else None
            h, c = h0.copy(), c0.copy()
            for step in range(gc.LM_STEPS):
                pred, y, h, c, _, _ = lm.predict(idx[step], h, c, vocab)
            key = "%s/%s/R%d" % (name, kind, rows)
            np.testing.assert_allclose(h, golden_lm[key + "/h"], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(c, golden_lm[key + "/c"], rtol=1e-12, atol=1e-14)
            ysel = y if kind == "subset" else y[:, cols]
            psel = pred if kind == "subset" else pred[:, cols]
            np.testing.assert_allclose(ysel, golden_lm[key + "/y"], rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(psel, golden_lm[key + "/pred"], rtol=1e-10, atol=1e-16)
            np.testing.assert_allclose(np.amax(y, axis=1), golden_lm[key + "/ymax"], rtol=1e-11, atol=1e-13)


def _run_case(case, fx, golden_decode, limit=None):
    name, fixture, kind, kwargs, spec = case
    f = fx(fixture)
    dec = (orc.OracleDynamicDecoder if kind == "dynamic" else orc.OracleDecoder)(f["root"], 1)
    sents = gc.case_sentences(spec, f["alphabet"])
    gold = golden_decode[name]
    assert [g["input"] for g in gold] == sents
    for si, s in enumerate(sents[:limit]):
        if kwargs.get("random_sampling"):
            np.random.seed(gc.RANDOM_SAMPLING_SEED + si)
        out = dec.decode(s, **kwargs)
        g = gold[si]["nbest"]
        assert len(out) == len(g)
        assert [w for _, w in out] == [w for _, w in g], (name, si)
        np.testing.assert_allclose([sc for sc, _ in out], [sc for sc, _ in g], rtol=1e-10, atol=1e-10)
        if "trace" in gold[si]:
            tr = dec.last_trace
            ends = dec.backward_lookup
            assert len(tr) == len(gold[si]["trace"])
            for i, (scores, prevs, nodes) in enumerate(tr):
                gi = gold[si]["trace"][i]
                assert len(scores) == len(gi)
                np.testing.assert_allclose(scores, [x[0] for x in gi], rtol=1e-10, atol=1e-10)
                assert [ends[i][n][0] for n in nodes] == [x[1] for x in gi]
                assert [ends[i][n][2] for n in nodes] == [x[2] for x in gi]


@pytest.mark.parametrize("case", FAST_DECODE, ids=[c[0] for c in FAST_DECODE])
def test_decode_matches_reference(case, fx, golden_decode):
    _run_case(case, fx, golden_decode)


def test_decode_config0_sample_matches_reference(fx, golden_decode):
    """BASELINE.json configs[0] (V=50k tied, beam 10): the first sentences of the
    100-sentence golden set (the full set is replayed on the GPU box)."""
    case = [c for c in gc.DECODE_CASES if c[0] == "mid-tied/static"][0]
    _run_case(case, fx, golden_decode, limit=3)


@pytest.mark.parametrize("case", SHAPED_FULL, ids=[c[0] for c in SHAPED_FULL])
def test_decode_shaped_sample_matches_reference(case, fx, golden_decode):
    """The BASELINE-size models with trained-model-like statistics (peaked logits, heavy-tailed blocks): the first sentences of
    each 24-sentence golden set (the full sets are replayed on the GPU box)."""
    _run_case(case, fx, golden_decode, limit=5)


def test_dynamic_requires_vocab_select(fx):
    f = fx("small-tied")
    dec = orc.OracleDynamicDecoder(f["root"], 1)
    with pytest.raises(TypeError):
        dec.decode("アイウ", vocab_select=False)
