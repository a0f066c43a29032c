import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from collections import deque
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.lattice import BatchLattice
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); eng = dec._engine; dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
lat = BatchLattice(dec._builder, sents, 10)
def run(n):
    q = deque()
    for _ in range(n):
        q.append(eng.submit(lat, "static", topN=10))
        if len(q) > dec.pipeline_depth: eng.collect(q.popleft())
    while q: eng.collect(q.popleft())
def timed(n, tag):
    run(3); torch.cuda.synchronize(); t = time.perf_counter(); run(n); torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (tag, (time.perf_counter() - t) / n * 1e3))
dec.decode_batch(sents * 3, beam_width=10)
dec.decode_batch(sents * 9, beam_width=10)
torch.cuda.synchronize(); t = time.perf_counter(); dec.decode_batch(sents * 40, beam_width=10); torch.cuda.synchronize()
print("strings -> strings, 40 chunks: %.3f ms/step" % ((time.perf_counter() - t) / 40 * 1e3))
timed(40, "device loop, fresh")
timed(12, "device loop, 12 steps")
eng.keep_n_live = True
for _ in range(20): eng.decode(lat, "static", topN=10, timing=True)
eng.keep_n_live = False
torch.cuda.synchronize()
timed(24, "device loop after timed decodes")
timed(12, "device loop, 12 steps")
eng.n_streams = 1; timed(12, "one stream"); eng.n_streams = 2; eng._rr = 0
timed(24, "two streams again")
print("plans:", len(eng.plans), [p.key[:5] for p in eng.plans])
