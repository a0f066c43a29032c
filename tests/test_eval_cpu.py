"""CPU-only: the eval harness (jlm_amd/eval.py) and the reference's own eval.py
run UNCHANGED over compat/ reproduce the hit counts and log body captured from
the reference (tests/golden/eval.json).  Device kernels are doubled by
tests/fake_hip.py; the same harness is exercised on the GPU in test_gpu_eval.py."""
import contextlib
import io
import os

import pytest

from jlm_amd import config as jconfig, synth
from tests import fake_hip
from tests import golden_cases as gc

REF_EVAL = "/root/reference/decoder/eval.py"


def _prepare(fx, fixture):
    f = fx(fixture)
    synth.write_test_corpus(f["root"], f["lexicon"], f["cfg"]["vocab_size"], **gc.EVAL_CORPUS)
    jconfig.set_root(f["root"])
    return f


def _body(path):
    with open(path, "r", encoding="utf-8") as fh:
        body = fh.read()
    return body[:body.index("--- ")] if "--- " in body else body


@pytest.mark.parametrize("case", gc.EVAL_CASES, ids=[c[0] for c in gc.EVAL_CASES])
@pytest.mark.parametrize("batch", [1, 8])
def test_eval_harness_matches_reference(case, batch, fx, golden_eval, monkeypatch, tmp_path):
    name, fixture, argv = case
    fake_hip.install(monkeypatch)
    _prepare(fx, fixture)
    monkeypatch.chdir(tmp_path)
    from jlm_amd import eval as jeval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        jeval.main(argv + ["--batch", str(batch)])
    gold = golden_eval[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    assert _body(os.path.join("eval", gold["log_name"])) == gold["log_body"]


@pytest.mark.skipif(not os.path.exists(REF_EVAL), reason="the reference is only present in the build container")
@pytest.mark.parametrize("case", gc.EVAL_CASES, ids=[c[0] for c in gc.EVAL_CASES])
def test_reference_eval_py_runs_unchanged_over_compat(case, fx, golden_eval, monkeypatch, tmp_path):
    name, fixture, argv = case
    fake_hip.install(monkeypatch)
    _prepare(fx, fixture)
    monkeypatch.chdir(tmp_path)
    os.makedirs("eval")
    from tools import run_reference_eval
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        run_reference_eval.run(REF_EVAL, argv)
    gold = golden_eval[name]
    assert [ln for ln in buf.getvalue().splitlines() if ln.startswith("best_hit")] == gold["stdout_hits"]
    assert _body(os.path.join("eval", gold["log_name"])) == gold["log_body"]
