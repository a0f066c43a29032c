"""GPU: the eval harness end to end on the HIP path reproduces the reference's
hit counts and log body (tests/golden/eval.json), sentence-at-a-time and batched."""
import contextlib
import io
import os

import pytest

torch = pytest.importorskip("torch")
from jlm_amd import config as jconfig, synth   # noqa: E402
from tests import golden_cases as gc           # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gc.EVAL_CASES, ids=[c[0] for c in gc.EVAL_CASES])
@pytest.mark.parametrize("batch", [1, 16])
def test_eval_harness_on_gpu(case, batch, fx, golden_eval, monkeypatch, tmp_path):
    name, fixture, argv = case
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    monkeypatch.chdir(tmp_path)
    from jlm_amd import eval as jeval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        jeval.main(argv + ["--batch", str(batch)])
    gold = golden_eval[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    with open(os.path.join("eval", gold["log_name"]), "r", encoding="utf-8") as fh:
        body = fh.read()
    assert body[:body.index("--- ")] == gold["log_body"]


def test_model_module_smoke_main(fx, capsys):
    """python -m jlm_amd.model: the reference module's __main__ smoke (model.py:213-245) on the HIP path; the per-word
    -log p it prints equal the oracle's for the sampled sequence."""
    import numpy as np
    from jlm_amd import model as jm
    from oracle import jlm_oracle as orc
    f = fx("small-vtable")
    np.random.seed(3)
    a, b = jm.main(["-e", "1", "--root", f["root"], "--steps", "10"])
    out = capsys.readouterr().out
    words = out.split("--- generated sentence\n")[1].splitlines()[0].split(" ")
    assert len(words) == 10 and np.isfinite(a) and np.isfinite(b)
    jconfig.set_root(f["root"])
    lm, o = jm.LSTM_Model(1), orc.OracleDecoder(f["root"], 1).model
    ids = np.random.RandomState(0).randint(1, f["cfg"]["vocab_size"], size=6).tolist()
    got = lm.evaluate(ids[0], ids[1:])
    # the evidently intended computation of model.py:200-206, on the oracle's explicit-state step
    h, c = o.zero_state(1)
    want, pred = [], None
    for i, w in enumerate(ids):
        if i:
            want.append(-np.log(pred[0, w]))
        pred, _y, h, c, _t1, _t2 = o.predict([w], h, c)
    np.testing.assert_allclose(got, want, rtol=1e-4)
