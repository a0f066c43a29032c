"""Sentence sharding across the GPUs of a node.

Sentences are independent (read-only weights and lexicon are replicated), so
the decode has NO collective on its data path: rank r decodes its own slice on
its own GPU.  Sentences are dealt round-robin in order of decreasing length so
every rank gets the same mix of lengths (the frame loop of a batch runs to its
longest sentence).  Only the optional result gather (Python n-best lists to
rank 0) touches torch.distributed, as an object gather on the control plane.
"""


def shard_indices(lengths, rank, world):
    """Indices of the sentences rank `rank` decodes (original order restored by
    :func:`merge`)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    return order[rank::world]


def decode_sharded(decoder, sentences, rank, world, **decode_kwargs):
    """-> (indices, results) for this rank's shard."""
    idx = shard_indices([len(s) for s in sentences], rank, world)
    if not idx:
        return idx, []
    return idx, decoder.decode_batch([sentences[i] for i in idx], **decode_kwargs)


def merge(n, parts):
    """parts: iterable of (indices, results) from every rank -> results in input order."""
    out = [None] * n
    for idx, res in parts:
        for i, r in zip(idx, res):
            out[i] = r
    return out


def gather_to_rank0(idx, res, n, dist, rank, world):
    """Control-plane gather of the n-best lists (not on the timed data path)."""
    parts = [None] * world if rank == 0 else None
    dist.gather_object((idx, res), parts, dst=0)
    return merge(n, parts) if rank == 0 else None
