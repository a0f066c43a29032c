"""What the GPU box gives the host side: visible CPUs, cgroup quota, and how the native lattice builder scales
with its thread count (one build at a time, then four concurrent builds as decode_batch's prefetch does)."""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
print("loadavg", open("/proc/loadavg").read().strip())
from jlm_amd import config as jconfig, synth
from jlm_amd.lattice import BatchLattice, LatticeBuilder
from jlm_amd.data import Vocab
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, lex, rd, al = synth.build_fixture(root, "mid-tied"); jconfig.set_root(root)
v = Vocab(cfg['vocab_size']); b = LatticeBuilder(lex, rd, v.w2i)
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
def build(dyn):
    lat = BatchLattice(b, sents, 10)
    if dyn:
        lat.dynamic_vocab(0, False, False, len(v.w2i))
for nt in (1, 2, 4, 8, 16, 32):
    b.n_threads = nt
    for dyn in (False, True):
        build(dyn)
        t = time.perf_counter()
        for _ in range(10): build(dyn)
        one = (time.perf_counter() - t) / 10 * 1e3
        def work():
            for _ in range(10): build(dyn)
        th = [threading.Thread(target=work) for _ in range(4)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        four = (time.perf_counter() - t) / 40 * 1e3
        print("threads %2d  %-8s one at a time %.2f ms   4 concurrent: %.2f ms per lattice" % (nt, "dynamic" if dyn else "static", one, four))
