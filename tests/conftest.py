import json
import os
import sys
import tempfile

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from jlm_amd import synth  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_FX = {}


def fixture_root(name):
    """Seeded synthetic artefacts for fixture ``name`` (cached per process)."""
    if name not in _FX:
        base = os.path.join(tempfile.gettempdir(), "jlm_test_fx_%d" % os.getuid())
        d = os.path.join(base, name)
        os.makedirs(d, exist_ok=True)
        cfg, lexicon, reading_dict, alphabet = synth.build_fixture(d, name)
        _FX[name] = dict(root=d, cfg=cfg, lexicon=lexicon, reading_dict=reading_dict, alphabet=alphabet)
    return _FX[name]


@pytest.fixture(scope="session")
def fx():
    return fixture_root


@pytest.fixture(scope="session")
def golden_decode():
    with open(os.path.join(GOLD, "decode.json"), "r", encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_lm():
    return np.load(os.path.join(GOLD, "lm_steps.npz"))


@pytest.fixture(scope="session")
def golden_eval():
    with open(os.path.join(GOLD, "eval.json"), "r", encoding="utf-8") as f:
        return json.load(f)
