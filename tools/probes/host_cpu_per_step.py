"""CPU seconds (all threads) the strings -> strings pipeline burns per 256-sentence step, and the step time when the process is
confined to N CPUs (taskset) -- what a rank gets when 8 ranks share a 16-CPU quota.  usage: host_cpu_per_step.py [ncpu ...]"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ncpus = [int(x) for x in sys.argv[1:]] or [0]
if len(ncpus) > 1 or ncpus[0]:
    import subprocess
    for n in ncpus:
        env = dict(os.environ, LOCAL_WORLD_SIZE=str(max(1, 16 // n)) if n else "1")
        cmd = ([] if not n else ["taskset", "-c", "0-%d" % (n - 1)]) + [sys.executable, __file__]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print("CPUs %2s: %s" % (n or "all", out[-1] if out else "?"))
    sys.exit(0)
import numpy as np, torch, jlm_amd
from jlm_amd import config as jconfig, synth
from jlm_amd.decoder import Decoder
root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
jconfig.set_root(root)
dec = Decoder(1); dec.max_batch = 256
sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
dec.decode_batch(sents * 8, beam_width=10)
N = 40
for blocking in (True, False, True, False):
  dec._engine.blocking_sync = blocking
  best = None
  for _ in range(3):
    torch.cuda.synchronize(); c0 = time.process_time(); t0 = time.perf_counter()
    dec.decode_batch(sents * N, beam_width=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; dc = time.process_time() - c0
    if best is None or dt < best[0]:
        best = (dt, dc)
  print("blocking_sync %s: %.3f ms/step wall, %.3f ms CPU per step (%.2f CPUs busy)" % (blocking, best[0] / N * 1e3, best[1] / N * 1e3, best[1] / best[0]))
best = None
for _ in range(3):
    torch.cuda.synchronize(); c0 = time.process_time(); t0 = time.perf_counter()
    dec.decode_batch(sents * N, beam_width=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; dc = time.process_time() - c0
    if best is None or dt < best[0]:
        best = (dt, dc)
print("usable_cpus %d, prefetch workers %d, lattice threads %d: %.3f ms/step wall, %.3f ms CPU per step (%.2f CPUs busy)" % (
    jlm_amd.usable_cpus(), dec.prefetch_workers, dec._builder.n_threads, best[0] / N * 1e3, best[1] / N * 1e3, best[1] / best[0]))
