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
    that measured them: it is replaced by this process's ``device_index``.  Rank ``src`` WINS: an entry this rank already
    holds for the same shape (an earlier local tuning, an ND_TUNE_CACHE file) is overwritten, and the number of such
    disagreements is reported.  Returns (choices taken over, of which replaced a different local choice)."""
    import torch.distributed as dist
    from . import _engine
    rank = dist.get_rank(group)
    payload = [None]
    if rank == src:
        payload[0] = [(list(k[1:]), list(v)) for k, v in _engine._TUNED.items()]
    dist.broadcast_object_list(payload, src=src, group=group)
    taken = replaced = 0
    if rank != src:
        for k, v in payload[0]:
            key = (device_index,) + tuple(tuple(e) if isinstance(e, list) else e for e in k)
            have = _engine._TUNED.get(key)
            if have != tuple(v):
                taken += 1
                replaced += have is not None
                _engine._TUNED[key] = tuple(v)
    return taken, replaced


def tune_on_rank0(model, batches, own=None, group=None):
    """Build (and thereby tune) the launch plans for ``batches`` -- the forward batch of EVERY rank's shard, i.e. rows per
    rank times 2 under classifier-free guidance; an int or an iterable of ints -- on rank 0 first, hand its choices to
    the other ranks, then let them build theirs from the shared choices.  ``own``: the forward batch rank 0 runs itself;
    plans it built only for the other ranks' sizes (ragged shards) are dropped again, since each holds a private copy of
    the packed weights.  A plan a rank built BEFORE the hand-over keeps the kernels it was built with, so call this
    before the first forward.  No-op without an initialised process group; with a world of one it still runs the
    broadcast (so a one-GPU box exercises the RCCL path)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 0
    sizes = sorted({int(batches)} if isinstance(batches, int) else {int(b) for b in batches})
    dev = next(model.parameters()).device
    if dist.get_rank(group) == 0:
        had = set(model._plans)
        for b in sizes:
            model._plan(b)
            key = (b, model.compute_dtype)
            if own is not None and b != own and key not in had:
                model._plans.pop(key, None)
    return share_tuned_choices(dev.index if dev.index is not None else 0, group)


def device_identity(device=None):
    """What tells one GPU from another in a bench line: ordinal, marketing name, architecture, UUID and PCI address as
    the runtime reports them (fields the installed torch does not expose are left out)."""
    import socket
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != 'cuda':
        return {'host': socket.gethostname(), 'device': str(dev)}
    p = torch.cuda.get_device_properties(dev)
    out = {'host': socket.gethostname(), 'device': dev.index, 'name': p.name}
    for k in ('gcnArchName', 'uuid', 'pci_domain_id', 'pci_bus_id', 'pci_device_id', 'multi_processor_count'):
        v = getattr(p, k, None)
        if v is not None:
            out[k] = v if isinstance(v, (int, str)) else str(v)
    return out
