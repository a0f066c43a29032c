"""CPU time per thread of the process over N pipelined steps (strings -> strings): who burns the CPUs -- the calling thread, the
lattice workers, the native builder's threads, the HIP runtime's."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open("/proc/self/task/%s/stat" % tid) as f:
                st = f.read()
            name = st[st.index("(") + 1:st.rindex(")")]
            fields = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(fields[11]) + int(fields[12])) / os.sysconf("SC_CLK_TCK"))
        except (OSError, ValueError):
            pass
    return out


root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.max_batch = 256
if len(sys.argv) > 2:
    dec.prefetch_workers, dec._builder.n_threads = int(sys.argv[1]), int(sys.argv[2])
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 8, beam_width=10)
N = 200
a = threads(); t0 = time.perf_counter()
dec.decode_batch(sents * N, beam_width=10)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
b = threads()
print("workers %d, lattice threads %d: %.3f ms/step wall" % (dec.prefetch_workers, dec._builder.n_threads, dt / N * 1e3))
rows = sorted(((b[t][1] - a.get(t, (None, 0.0))[1], b[t][0], t) for t in b), reverse=True)
for cpu, name, tid in rows[:14]:
    print("  %-18s tid %-8d %7.3f ms CPU per step (%.0f %% of a CPU)%s" % (name, tid, cpu / N * 1e3, 100 * cpu / dt, "  <- calling thread" if tid == os.getpid() else ""))
print("  total %.3f ms CPU per step" % (sum(r[0] for r in rows) / N * 1e3))
