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


def share_tuned_choices(device_index, group=None, src=0):
    """Rank ``src``'s measured kernel choices (``_engine._TUNED``) become every rank's.  Each process would otherwise time
    its tile variants on its own GPU and ranks could settle on different -- individually bit-identical within a kernel
    family, but not across families -- kernels for the same shape, so a sharded run would not be comparable bit for bit
    with the single-process one; tuning once also saves N - 1 tuning runs.  Keys carry the device ordinal of the process
    that measured them: it is replaced by this process's ``device_index``.  Returns the number of choices taken over."""
    import torch.distributed as dist
    from . import _engine
    rank = dist.get_rank(group)
    payload = [None]
    if rank == src:
        payload[0] = [(list(k[1:]), list(v)) for k, v in _engine._TUNED.items()]
    dist.broadcast_object_list(payload, src=src, group=group)
    taken = 0
    if rank != src:
        for k, v in payload[0]:
            key = (device_index,) + tuple(tuple(e) if isinstance(e, list) else e for e in k)
            if key not in _engine._TUNED:
                _engine._TUNED[key] = tuple(v)
                taken += 1
    return taken


def tune_on_rank0(model, batch, group=None):
    """Build (and thereby tune) the launch plan for ``batch`` images per rank on rank 0 first, hand its choices to the
    other ranks, then let them build theirs from the shared choices.  No-op without torch.distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    dev = next(model.parameters()).device
    if dist.get_rank(group) == 0:
        model._plan(batch)
    return share_tuned_choices(dev.index if dev.index is not None else 0, group)
