"""Race hunt: many ragged chunks through the pipelined product path (one batch in flight per launch stream, page-locked lattice blocks reused, frame-loop op,
lattice prefetch threads, plans reused while others are in flight) against the SAME device batches decoded one at a time on one
stream, timed.  Same kernels, same operands: every n-best list and every score must be bit-identical."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.decoder_dynamic import DynamicDecoder
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
bad = 0
for fixture, cls, kw in (("mid-vtable", Decoder, {}), ("mid-tied", Decoder, dict(vocab_select=True)),
                         ("mid-tied", DynamicDecoder, dict(vocab_select=True)), ("mid-tied", Decoder, {})):
    cfg, _l, _r, al = synth.build_fixture(root, fixture)
    jconfig.set_root(root)
    dec = cls(1)
    dec.perf_timing = False
    dec.max_batch = 192
    sents = synth.make_ragged_sentences(192 * chunks, 1, 30, seed=99, alphabet=al)
    # (rounds 2-3: a CU share below 100 % changed the vocabulary kernel's column cuts -- the summation order of the last float32 bits --
    #  whenever another batch was in flight, and this probe switched the share off.  Round 4: the default share is 100, both runs cut alike)
    if os.environ.get("SOAK_SHARE"):
        dec._engine.lse_share_pct = int(os.environ["SOAK_SHARE"])
    t = time.perf_counter()
    fast = dec.decode_batch(sents, beam_width=10, **kw)
    t_fast = time.perf_counter() - t
    eng = dec._engine
    batches = dec._chunks(sents, 10)         # the same device batches (dealt by decreasing length), one at a time on one stream
    eng.n_streams, dec.perf_timing, dec.pipeline_depth, dec.prefetch_workers = 1, True, 0, 1
    t = time.perf_counter()
    slow = [None] * len(sents)
    for idx in batches:
        for j, r in zip(idx, dec.decode_batch([sents[j] for j in idx], beam_width=10, **kw)):
            slow[j] = r
    t_slow = time.perf_counter() - t
    n_diff = sum(1 for a, b in zip(fast, slow) if a != b)
    bad += n_diff
    print("%-11s %-15s %-22s %6d sentences: %d differ   pipelined %.2f s, serial %.2f s" % (
        fixture, cls.__name__, kw, len(sents), n_diff, t_fast, t_slow))
    if n_diff:
        for i, (a, b) in enumerate(zip(fast, slow)):
            if a != b:
                print("  first difference at sentence", i, sents[i], a[:2], b[:2]); break
    del dec
    torch.cuda.empty_cache()
print("TOTAL differing sentences:", bad)
sys.exit(1 if bad else 0)
