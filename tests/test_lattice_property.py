"""CPU-only, property-based: the native lattice builder (libjlm_host.so) = the Python builder = the oracle's restatement of the reference's
_build_lattice (decoder/decoder.py:79-135) on RANDOM lexicons -- duplicate readings, readings that are prefixes of one another, words past
the vocabulary size (skipped, decoder.py:99-103), frames nothing ends in (the raw-symbol <unk> fallback, :128-130), empty and one-symbol
sentences -- not only on the seeded fixtures."""
import numpy as np
import pytest

hyp = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st, HealthCheck        # noqa: E402

from jlm_amd import lattice                                                 # noqa: E402
from jlm_amd.data import Vocab                                              # noqa: E402
from oracle import jlm_oracle as orc                                        # noqa: E402

ARRAYS = ("node_start", "node_word", "node_lex", "node_sent", "node_end", "end_off", "sg_off", "sg_node", "sg_word")
ALPHA = "ァアィイゥ"
reading = st.text(alphabet=ALPHA, min_size=1, max_size=4)
sentence = st.text(alphabet=ALPHA + "ヷ", min_size=0, max_size=12)          # ヷ: in no reading -> <unk> fallback nodes


def _lexicon(readings):
    """the reference's layout: (word, freq) sorted by (-freq, word), <eos> first; reading_dict reading -> lexicon indices"""
    n = len(readings)
    lex = [("<eos>", n + 11)] + [("w%d/%s/N" % (i, r), n + 10 - i) for i, r in enumerate(readings)]
    rd = {}
    for i, (w, _f) in enumerate(lex):
        t = w.split("/")
        if len(t) >= 3:
            rd.setdefault(t[1], []).append(i)
    return lex, rd


@pytest.mark.skipif(lattice.host_lib() is None, reason="libjlm_host.so not built")
@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(readings=st.lists(reading, min_size=1, max_size=40), oov=st.integers(min_value=0, max_value=6),
       sents=st.lists(sentence, min_size=1, max_size=6), beam=st.integers(min_value=1, max_value=5))
def test_native_python_and_oracle_agree_on_random_lexicons(readings, oov, sents, beam):
    lex, rd = _lexicon(readings)
    vsize = max(2, len(lex) - oov)                 # Vocab keeps the first size - 1 lexicon entries: the tail is out of vocabulary
    v = Vocab(vsize, lex)
    b = lattice.LatticeBuilder(lex, rd, v.w2i)
    b.use_native = False
    l0 = lattice.BatchLattice(b, sents, beam)
    b.use_native = True
    l1 = lattice.BatchLattice(b, sents, beam)
    assert (l1.n_nodes, l1.max_cands, l1.n_frames) == (l0.n_nodes, l0.max_cands, l0.n_frames)
    for k in ARRAYS:
        np.testing.assert_array_equal(getattr(l0, k), getattr(l1, k), err_msg=k)
    for s, text in enumerate(sents):
        assert l1.backward_lookup(s) == orc.build_lattice(text, lex, rd, v.w2i), (s, text)
    w0, o0, _ = l0.static_vocab()
    w1, o1, _ = l1.static_vocab()
    np.testing.assert_array_equal(w0, w1)
    np.testing.assert_array_equal(o0, o1)
