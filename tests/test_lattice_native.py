"""CPU-only: the native lattice builder (libjlm_host.so, include/jlm_host.h) against
the pure-Python builder (the specification) and against the oracle's restatement of
the reference's _build_lattice, on ragged batches with OOV words and raw-symbol
fallbacks."""
import numpy as np
import pytest

from jlm_amd import lattice, synth
from jlm_amd.data import Vocab
from oracle import jlm_oracle as orc

ARRAYS = ("node_start", "node_word", "node_lex", "node_sent", "node_end", "end_off", "sg_off", "sg_node", "sg_word")


def _builder(f):
    v = Vocab(f["cfg"]["vocab_size"], f["lexicon"])
    return lattice.LatticeBuilder(f["lexicon"], f["reading_dict"], v.w2i), v


@pytest.mark.skipif(lattice.host_lib() is None, reason="libjlm_host.so not built")
@pytest.mark.parametrize("fixture,n,lo,hi,alpha", [("small-tied", 40, 1, 25, 12), ("small-tied", 7, 1, 3, 80),
                                                     ("mid-tied", 64, 5, 30, 80)])
def test_native_equals_python(fixture, n, lo, hi, alpha, fx):
    f = fx(fixture)
    b, v = _builder(f)
    sents = synth.make_ragged_sentences(n, lo, hi, seed=n + hi, alphabet=alpha)    # alpha=80 on the 12-kana lexicon: unk fallbacks
    b.use_native = False
    l0 = lattice.BatchLattice(b, sents, 7)
    b.use_native = True
    l1 = lattice.BatchLattice(b, sents, 7)
    assert l1.n_nodes == l0.n_nodes and l1.max_cands == l0.max_cands and l1.n_frames == l0.n_frames
    for k in ARRAYS:
        np.testing.assert_array_equal(getattr(l0, k), getattr(l1, k), err_msg=k)
    for kw in (dict(), dict(samples=9, top_sampling=True)):
        b.use_native = False
        w0, o0, ls0 = l0.static_vocab(**kw)
        b.use_native = True
        w1, o1, ls1 = l1.static_vocab(**kw)
        np.testing.assert_array_equal(w0, w1)
        np.testing.assert_array_equal(o0, o1)
        assert ls1[-1] == ls0[-1] and ls1[0] == ls0[0]
    # against the oracle's restatement of the reference, sentence by sentence
    for s in (0, n // 2, n - 1):
        assert l1.backward_lookup(s) == orc.build_lattice(sents[s], f["lexicon"], f["reading_dict"], v.w2i)


@pytest.mark.skipif(lattice.host_lib() is None, reason="libjlm_host.so not built")
def test_native_retry_when_capacity_is_short(fx, monkeypatch):
    f = fx("small-tied")
    b, _ = _builder(f)
    sents = ["ァィ" * 40] * 3               # long, dense sentences: node count far above the first guess
    b.use_native = False
    l0 = lattice.BatchLattice(b, sents, 3)
    b.use_native = True
    l1 = lattice.BatchLattice(b, sents, 3)
    np.testing.assert_array_equal(l0.node_word, l1.node_word)


@pytest.mark.skipif(lattice.host_lib() is None, reason="libjlm_host.so not built")
@pytest.mark.parametrize("kw", [dict(), dict(samples=6, top_sampling=True), dict(samples=5, random_sampling=True)])
def test_native_dynamic_vocab_lists(kw, fx):
    f = fx("small-tied")
    b, v = _builder(f)
    sents = synth.make_ragged_sentences(23, 1, 18, seed=77, alphabet=f["alphabet"])
    outs = []
    for native in (False, True):
        b.use_native = native
        lat = lattice.BatchLattice(b, sents, 5)
        np.random.seed(11)
        outs.append(lat.dynamic_vocab(vocab_len=len(v.w2i), **kw))
    for a, c in zip(outs[0][:4], outs[1][:4]):
        np.testing.assert_array_equal(a, c)
    assert outs[0][4][-1] == outs[1][4][-1]
    # the init list of every cell is, as a multiset, the reference's lv[k] + delta[k+1] (decoder_dynamic.py:30-46,112-127)
    lat = lattice.BatchLattice(b, sents, 5)
    np.random.seed(11)
    extra = None
    if kw.get("random_sampling"):
        extra = [[int(x) for x in np.random.randint(len(v.w2i), size=kw["samples"])] for _ in sents]
    elif kw.get("top_sampling"):
        extra = [list(range(kw["samples"]))] * len(sents)
    for s in range(len(sents)):
        lv, d = lat._dyn_lists_python(s, extra[s] if extra else [])
        for k in range(len(sents[s])):
            assert sorted(lat.dynamic_init_list(outs[1], k, s)) == sorted(lv[k] + d[k + 1]), (s, k)
        assert lat.dynamic_init_list(outs[1], len(sents[s]), s) == []
    # the final per-frame lists of a sentence equal the reference's lattice_vocab after its decode
    if not kw.get("random_sampling"):
        o = orc.OracleDynamicDecoder(f["root"], 1)
        o.decode(sents[3], vocab_select=True, beam_width=5, **kw)
        assert outs[1][4][3] == o.lattice_vocab
