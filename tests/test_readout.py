"""CPU: the C read-out (jlm_amd._readout.nbest, csrc/jlm_readout.c) builds exactly the lists the numpy
implementation does -- ragged sentence lengths, empty ranks, topN < beam, <unk> fallback nodes."""
import numpy as np
import pytest

from jlm_amd import engine
from jlm_amd.lattice import BatchLattice, LatticeBuilder


def _lattice(n_sent, beam, seed):
    rng = np.random.RandomState(seed)
    kana = [chr(0x30A1 + i) for i in range(12)]
    lexicon = [("<eos>", 10 ** 6)]
    reading_dict = {}
    for i in range(60):
        r = "".join(rng.choice(kana[:8], size=rng.randint(1, 4)))      # kana[8:] never has a word: <unk> fallback
        lexicon.append(("w%d/%s/N" % (i, r), 1000 - i))
        reading_dict.setdefault(r, []).append(len(lexicon) - 1)
    w2i = {"<unk>": 0}
    for j, (w, _f) in enumerate(lexicon):
        w2i[w] = j + 1
    b = LatticeBuilder(lexicon, reading_dict, w2i)
    texts = ["".join(rng.choice(kana, size=rng.randint(1, 15))) for _ in range(n_sent)]
    return BatchLattice(b, texts, beam), rng


@pytest.mark.parametrize("n_sent,beam,top", [(1, 1, 10), (7, 4, 10), (33, 10, 3), (16, 6, 6)])
def test_c_readout_equals_numpy_readout(n_sent, beam, top):
    ext = engine._readout_ext()
    assert ext is not None, "jlm_amd/_readout.so not built (python __graft_entry__.py)"
    lat, rng = _lattice(n_sent, beam, 5 + n_sent)
    assert (lat.node_lex < -1).any() or n_sent == 1          # the fallback path is exercised
    rmax, stride = n_sent * beam, lat.n_frames + 1
    nodes = np.zeros((rmax, stride), dtype=np.int32)
    lens = np.zeros(rmax, dtype=np.int32)
    score = rng.rand(rmax)
    for s in range(n_sent):
        n_rank = rng.randint(0, beam + 1)                      # some sentences have fewer paths than beam, some none
        for r in range(n_rank):
            k = rng.randint(0, lat.n_frames)                   # words on the path (0: only the root)
            ids = rng.randint(0, lat.n_nodes, size=k + 1)
            nodes[s * beam + r, :k + 1] = ids
            lens[s * beam + r] = k + 1
        if n_rank + 1 < beam:                                  # a filled rank after an empty one must be ignored
            lens[s * beam + n_rank + 1] = 2
    a = engine.DecodeEngine._read_out(lat, nodes, lens, score, top)
    b = engine.DecodeEngine._read_out_py(lat, nodes, lens, score, top)
    assert a == b
    assert len(a) == n_sent and all(len(x) <= min(top, beam) for x in a)


def test_c_readout_rejects_short_arrays():
    ext = engine._readout_ext()
    lat, _ = _lattice(4, 3, 1)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    with pytest.raises(ValueError):
        ext.nbest(np.zeros((5, 8), np.int32), np.zeros(12, np.int32), np.zeros(12), i32(lat.node_lex), i32(lat.node_sent),
                  i32(lat.node_start), lat.builder.lex_list, lat.texts, 4, 3, 10, 8)
    with pytest.raises(TypeError):
        ext.nbest(np.zeros((12, 8), np.int64), np.zeros(12, np.int32), np.zeros(12), i32(lat.node_lex), i32(lat.node_sent),
                  i32(lat.node_start), lat.builder.lex_list, lat.texts, 4, 3, 10, 8)
