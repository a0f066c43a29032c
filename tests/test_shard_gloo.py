"""CPU-only, world_size 2 over gloo: the sentence-sharded decode
returns exactly the single-process result.  Device kernels are doubled by
tests/fake_hip.py inside each spawned rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jlm_amd import shard
from tests import golden_cases as gc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, root, sents, kwargs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jlm_amd import config as jconfig
        from tests import fake_hip
        fake_hip.install_plain()
        jconfig.set_root(root)
        from jlm_amd.decoder import Decoder
        dec = Decoder(1)
        dec.perf_timing = False
        idx, res = shard.decode_sharded(dec, sents, rank, world, **kwargs)
        dist.barrier()
        merged = shard.gather_to_rank0(idx, res, len(sents), dist, rank, world)
        if rank == 0:
            q.put(merged)
    finally:
        dist.destroy_process_group()


def test_shard_indices_cover_everything_once():
    lens = [5, 20, 1, 7, 7, 13, 2, 9, 11]
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in shard.shard_indices(lens, r, world))
        assert seen == list(range(len(lens)))
    # longest sentences are spread over the ranks
    assert {shard.shard_indices(lens, r, 2)[0] for r in range(2)} == {1, 5}


def test_two_rank_sharded_decode_equals_golden(fx, golden_decode):
    case = [c for c in gc.DECODE_CASES if c[0] == "small-vtable/static"][0]
    name, fixture, _kind, kwargs, spec = case
    f = fx(fixture)
    sents = gc.case_sentences(spec, f["alphabet"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, f["root"], sents, kwargs, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gold = golden_decode[name]
    assert len(merged) == len(sents)
    for si, out in enumerate(merged):
        assert [w for _, w in out] == [w for _, w in gold[si]["nbest"]]
