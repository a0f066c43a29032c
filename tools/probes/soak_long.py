"""Long run: decode_batch over many ragged batches; device memory, host RSS and throughput must stay flat."""
import os, sys, tempfile, time, resource
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
from jlm_amd.decoder_dynamic import DynamicDecoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-tied")
jconfig.set_root(root)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 40
for cls, kw in ((Decoder, {}), (DynamicDecoder, dict(vocab_select=True))):
    dec = cls(1); dec.perf_timing = False; dec.max_batch = 256
    t0 = time.time(); n = 0; it = 0
    marks = []
    while time.time() - t0 < secs:
        sents = synth.make_ragged_sentences(256 * 8, 1, 40, seed=1000 + it, alphabet=al)
        out = dec.decode_batch(sents, beam_width=10, **kw)
        assert len(out) == len(sents) and all(o for o in out)
        n += sum(len(s) for s in sents); it += 1
        if it % 10 == 1:
            marks.append((it, torch.cuda.memory_allocated() >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, len(dec._engine.plans)))
    dt = time.time() - t0
    print("%-15s %d calls, %.0f chars/s; (call, device MB, host RSS MB, plans): %s" % (cls.__name__, it, n / dt, [marks[i] for i in sorted(set([0, len(marks) // 4, len(marks) // 2, 3 * len(marks) // 4, len(marks) - 1]))]))
    del dec; torch.cuda.empty_cache()
