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
    # The built libraries are git-ignored; on a tree that has none yet, build them once (hipcc cross-compiles
    # gfx950 without a GPU).  An existing build is left alone: rebuilding is __graft_entry__.build()'s job.
    built = [os.path.join(REPO, "jlm_amd", "csrc", "libjlm_hip.so"), os.path.join(REPO, "jlm_amd", "csrc", "libjlm_host.so"),
             os.path.join(REPO, "jlm_amd", "_readout.so"), os.path.join(REPO, "jlm_amd", "_torch_ops.so")]
    if not all(os.path.exists(b) for b in built):
        try:
            import __graft_entry__ as ge
            ge.build()
        except Exception as e:          # the tests that need a library say so themselves
            print("conftest: build() failed: %r" % (e,))


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


@pytest.fixture(scope="session")
def golden_char():
    with open(os.path.join(GOLD, "char.json"), "r", encoding="utf-8") as f:
        return json.load(f)
