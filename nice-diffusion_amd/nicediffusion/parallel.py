"""Batch sharding of the sampling loop over the GPUs of one node (one process per GPU, torch.distributed).

The reference has no distributed code (trainer.py:9 lists it as a TODO).  The path shards trivially along the batch
dimension: GroupNorm statistics are per (sample, group), attention is per sample and the sampler update is
elementwise, so ranks never talk inside the loop; the only exchange is one all-gather of the finished samples
(backend 'nccl' = RCCL over xGMI on AMD GPUs; 'gloo' in the CPU tests).
"""
import torch


def shard_slice(n, rank, world):
    """Rows [lo, hi) of an n-row global batch owned by ``rank``; the first n % world ranks get one extra row."""
    if not 0 <= rank < world:
        raise ValueError('rank {} outside world of {}'.format(rank, world))
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return slice(lo, lo + base + (1 if rank < extra else 0))


def all_gather_rows(local, n, rank, world, group=None):
    """Concatenate every rank's rows (in rank order) into the global [n, ...] tensor on every rank."""
    import torch.distributed as dist
    sizes = [shard_slice(n, r, world) for r in range(world)]
    counts = [s.stop - s.start for s in sizes]
    assert local.shape[0] == counts[rank], 'local rows {} != shard size {}'.format(local.shape[0], counts[rank])
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == 'gloo':
        # rehearsal / test configuration only: stage through host memory (gloo has no device all-gather)
        return all_gather_rows(local.cpu(), n, rank, world, group).to(local.device)
    if len(set(counts)) == 1:
        out = torch.empty((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # ragged: pad every shard to the largest, gather, then trim
    m = max(counts)
    padded = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:counts[rank]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)
