"""CPU-only: every weight file format of the reference's tools (train/weights.py verbose
dumps, train/comp.py k-means files) loads to the same arrays, and the decoder runs on the
compressed model exactly like the oracle run on the decoded weights."""
import os
import shutil

import numpy as np
import pytest

from jlm_amd import config as jconfig, synth, weights as W
from oracle import jlm_oracle as orc
from tests import fake_hip


@pytest.fixture()
def exp(tmp_path):
    root = str(tmp_path)
    cfg = synth.make_config(600, 32, 16, "tied")
    synth.write_lexicon(root, 600, alphabet=8)
    synth.write_experiment(root, 1, cfg, scale=0.3)
    jconfig.set_root(root)
    return root, cfg


def test_compressed_formats_agree(exp):
    root, cfg = exp
    decoded = synth.write_compressed(root, 1, bit=8)
    d = W.weights_dir(1)
    a = W.load_weights(1, 8)                                        # decoded pickle (model.py:74-78)
    os.remove(os.path.join(d, "lstm_weights_comp_8.pkl"))
    b = W.load_weights(1, 8)                                        # (code, codebook) dump
    os.remove(os.path.join(d, "comp_8", "lstm_weights_comp_dump.pkl"))
    c = W.load_weights(1, 8)                                        # debug text files
    for k, v in decoded.items():
        np.testing.assert_array_equal(a[k], v)
        np.testing.assert_array_equal(b[k], v)
        np.testing.assert_allclose(c[k], v, rtol=1e-6)
        assert len(np.unique(v)) <= 256
    shutil.rmtree(os.path.join(d, "comp_8"))
    with pytest.raises(FileNotFoundError):
        W.load_weights(1, 8)


@pytest.mark.parametrize("mode", ["tied", "vtable", "dsoftmax"])
def test_verbose_dumps_load_without_the_pickle(mode, tmp_path):
    root = str(tmp_path)
    cfg = synth.make_config(400, 32, 16, mode, segs=[(16, 0, 100), (8, 100, 250), (4, 250, None)])
    synth.write_lexicon(root, 400, alphabet=8)
    ref = synth.write_experiment(root, 1, cfg, scale=0.3)
    jconfig.set_root(root)
    synth.write_verbose_dumps(root, 1, npy=(mode != "dsoftmax"))
    os.remove(os.path.join(W.weights_dir(1), "lstm_weights.pkl"))
    got = W.load_weights(1, 0, cfg)
    assert set(got) == set(ref)
    for k, v in ref.items():
        if isinstance(v, list):
            for x, y in zip(got[k], v):
                np.testing.assert_allclose(x, y, rtol=1e-6)
        else:
            np.testing.assert_allclose(got[k], v, rtol=1e-6)


def test_decoder_on_compressed_model_matches_oracle(exp, monkeypatch):
    root, cfg = exp
    synth.write_compressed(root, 1, bit=6, formats=("dump",))
    fake_hip.install(monkeypatch)
    from jlm_amd.decoder import Decoder
    dec = Decoder(1, comp=6)
    dec.perf_timing = False
    # the vocabulary block was expanded from the resident (code, codebook) pair by the device-side de-quantiser
    m = dec.model.dev
    assert 0 in m.seg_codes and m.seg_codes[0][0].dtype == __import__("torch").uint8
    np.testing.assert_array_equal(m.seg_B[0].numpy()[:, :m.segments[0]["k"]][:, :16], W.load_weights(1, 6)["LM"])
    # oracle on the decoded weights (what model.py:74-78 would have unpickled)
    decoded = W.load_weights(1, 6)
    o = orc.OracleDecoder(root, 1)
    o.model = orc.OracleLM(o.config, decoded)
    sents = synth.make_ragged_sentences(5, 2, 10, seed=2, alphabet=8)
    for s, g in zip(sents, dec.decode_batch(sents, beam_width=5)):
        want = o.decode(s, beam_width=5)
        assert [w for _, w in g] == [w for _, w in want]
        np.testing.assert_allclose([x for x, _ in g], [x for x, _ in want], rtol=1e-5, atol=1e-4)
