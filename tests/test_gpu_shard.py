"""GPU, BASELINE.json configs[4]: the sentence-sharded decode of the tied V=50k model.  Two ranks over gloo,
BOTH on cuda:0 (the GPU boxes have one device; the 8-GPU run is the driver's), each decoding its shard through
jlm_amd.shard.decode_sharded exactly as `bench.py --config 5` does; the merged n-best lists must equal the one-rank decode
of the same sentences -- the shards cover every sentence once, and a sentence's result does not depend on which other
sentences share its batch (beyond the last float32 bits of a score)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402

from jlm_amd import shard, synth          # noqa: E402

pytestmark = pytest.mark.gpu

N_SENT = 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, root, sents, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    import jlm_amd  # noqa: F401  (defaults GPU_MAX_HW_QUEUES before the HIP runtime starts)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jlm_amd import config as jconfig
        jconfig.set_root(root)
        from jlm_amd.decoder import Decoder
        dec = Decoder(1)
        dec.max_batch = 128                       # several pipelined chunks per rank
        idx, res = shard.decode_sharded(dec, sents, rank, world, beam_width=10)
        dist.barrier()
        merged = shard.gather_to_rank0(idx, res, len(sents), dist, rank, world)
        if rank == 0:
            q.put(merged)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank(fx):
    f = fx("mid-tied")
    sents = synth.make_sentences(N_SENT - 40, 20, seed=5555, alphabet=f["alphabet"]) + \
        synth.make_ragged_sentences(40, 3, 31, seed=9, alphabet=f["alphabet"])          # ragged lengths exercise the length-sorted deal
    lens = [len(s) for s in sents]
    a, b = shard.shard_indices(lens, 0, 2), shard.shard_indices(lens, 1, 2)
    assert sorted(a + b) == list(range(len(sents))) and abs(sum(lens[i] for i in a) - sum(lens[i] for i in b)) <= max(lens)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, f["root"], sents, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from jlm_amd import config as jconfig
    jconfig.set_root(f["root"])
    from jlm_amd.decoder import Decoder
    one = Decoder(1).decode_batch(sents, beam_width=10)
    assert len(merged) == len(one) == len(sents)
    # the two runs cut the batch differently (512- vs 1 024-sentence device batches: other vocabulary ranges, other
    # summation order in the last float32 bits), so the order may differ only between hypotheses that tie at 1e-6
    n_reordered = 0
    for si, (x, y) in enumerate(zip(merged, one)):
        assert len(x) == len(y), si
        assert x[0][1] == y[0][1], ("1-best differs", si)
        np.testing.assert_allclose([v for v, _ in x], [v for v, _ in y], rtol=0, atol=2e-5)
        if [w for _, w in x] == [w for _, w in y]:
            continue
        n_reordered += 1
        yscore = {tuple(w): v for v, w in y}
        for i, (_v, w) in enumerate(x):
            if w != y[i][1]:
                ref = yscore.get(tuple(w), y[-1][0])
                assert abs(ref - y[i][0]) <= 1e-6 * max(1.0, abs(y[i][0])), ("order differs beyond a tie", si, i, ref, y[i][0])
    assert n_reordered <= len(sents) // 50, n_reordered


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell (no launcher, WORLD_SIZE unset) starts the two ranks itself and prints ONE
    JSON line with n_gpus = 2 (--debug-shared-gpu: both ranks on this box's one device, control plane over gloo)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--debug-shared-gpu", "--steps", "3", "--warmup", "1",
                        "--no-config5", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["untimed_steps"] >= 1 and d["roofline"]["frac"] > 0
    # a world size that contradicts --gpus is refused with a message, not an assert
    r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1"],
                        env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r2.stderr + r2.stdout)
