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
