"""Host-path ceiling with N processes on one box (the verdict's 8-rank question): every process decodes 256-sentence batches strings -> strings on
device 0 with the frame loop's main kernels LEFT OUT (a -DJLM_PROBE_SKIP build swapped in for libjlm_hip.so, JLM_SKIP=23: results are wrong by
construction, the GPU does next to nothing), so the rate is what the host side -- lattice workers, staging, enqueue, read-out -- sustains under
the box's CPU quota.  usage: python tools/probes/host_soak8.py N [seconds]   (run by tools/probes/host_soak8.sh, which swaps the library)"""
import os, subprocess, sys, tempfile, time
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch, jlm_amd
    from jlm_amd import config as jconfig, synth
    from jlm_amd.decoder import Decoder
    secs, world = float(sys.argv[2]), int(sys.argv[3])
    os.environ["LOCAL_WORLD_SIZE"] = str(world)          # usable_cpus() divides the quota by the ranks of the node
    root = os.path.join(tempfile.gettempdir(), "jlm_dbg")
    cfg, _l, _r, al = synth.build_fixture(root, "mid-vtable")
    jconfig.set_root(root)
    dec = Decoder(1); dec.max_batch = 256
    sents = synth.make_sentences(256, 20, seed=4242, alphabet=al)
    dec.decode_batch(sents * 12, beam_width=10)
    t_end = time.time() + secs
    n = 0; c0 = time.process_time(); t0 = time.perf_counter()
    while time.time() < t_end:
        dec.decode_batch(sents * 20, beam_width=10)
        n += 20
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("RESULT %.3f ms/step  %.0f chars/s  %.2f CPUs  workers %d" % (dt / n * 1e3, n * 5120 / dt, (time.process_time() - c0) / dt, dec.prefetch_workers), flush=True)
    sys.exit(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
env = dict(os.environ, JLM_SKIP=os.environ.get("JLM_SKIP", "23"))
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(secs), str(N)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
      for _ in range(N)]
tot = 0.0
for i, p in enumerate(ps):
    out, err = p.communicate()
    line = [l for l in out.splitlines() if l.startswith("RESULT")]
    print("process %d: %s" % (i, line[0][7:] if line else "FAILED: " + " | ".join(err.strip().splitlines()[-3:])))
    if line:
        tot += float(line[0].split()[3])
print("%d processes: %.2f M chars/s in total (host path only; one GPU decodes 2.4-2.5 M chars/s per rank)" % (N, tot / 1e6))
