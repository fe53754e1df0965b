"""World-size-2 test of the batch-sharded sampling path on CPU (gloo).  The HIP forward is replaced by the CPU
oracle inside the worker (tests may use the oracle); what is under test is the product's shard/gather logic
(``Diffusion.denoise_sharded``, ``parallel.all_gather_rows``), i.e. the N>1 code path of bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nicediffusion.diffusion import Diffusion
    from nicediffusion.parallel import all_gather_rows, shard_slice
    from oracle import unet_oracle as UO, diffusion_oracle as DO
    from tests.cases import TINY_CFGS
    torch.set_num_threads(2)
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
    try:
        cfg = dict(TINY_CFGS['adagn_updown'])
        sd = UO.synth_state_dict(cfg, seed=99)
        so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, 4, 'cosine'),
                              'learned_interpolation', use_ddim=True, ddim_eta=0.0)

        class Stub:
            """Quacks like Diffusion for denoise_sharded; the per-rank denoise is the CPU oracle."""
            calls = []

            def denoise(self, x=None, kwargs=None, batch_size=1, noise=None, **kw):
                assert x.shape[0] == batch_size == len(kwargs['y'])
                Stub.calls.append(x.shape[0])
                return so.denoise(x, kwargs['y'])

        torch.manual_seed(0)
        x = torch.randn(n, 3, 16, 16)
        y = (torch.arange(n) * 37) % 10
        out = Diffusion.denoise_sharded(Stub(), x, kwargs={'y': y})
        sl = shard_slice(n, rank, world)
        assert Stub.calls == [sl.stop - sl.start]
        # ragged gather of a second tensor
        g = all_gather_rows(torch.full((sl.stop - sl.start, 2), float(rank)), n, rank, world)
        if rank == 0:
            full = so.denoise(x, y)
            q.put((out.numpy(), full.numpy(), g.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n,world', [(4, 2), (5, 2), (8, 8), (11, 8)])
def test_sharded_denoise_matches_single_process(n, world):
    """world 2 and world 8 (north_star's node: 8 ranks, one per GPU; here 8 gloo ranks on the CPU), even and ragged shards"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, full, g = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert out.shape == full.shape
    # rows are computed independently per sample: an N-rank run reproduces the 1-process run row by row
    assert np.abs(out - full).max() < 1e-5
    from nicediffusion.parallel import shard_slice
    exp = np.concatenate([np.full((shard_slice(n, r, world).stop - shard_slice(n, r, world).start, 2), float(r)) for r in range(world)])
    assert np.array_equal(g, exp)


def test_bench_multi_rank_branch_over_gloo():
    """bench.py's own N > 1 branch (shard_slice of the global batch, one_pass + all_gather_rows, barrier, MAX-reduced
    time, rank 0's JSON line) with world_size 2 over gloo and the denoiser stubbed out (ND_BENCH_STUB=1: no GPU work)."""
    import json
    import subprocess
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE='2', RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), ND_BENCH_BACKEND='gloo', ND_BENCH_STUB='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3',
                                       '--warmup', '1', '--batch', '5'], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert not [l for l in outs[1][0].splitlines() if l.startswith('{')], 'only rank 0 prints the JSON line'
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['warmup'] == 1 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 10 and rec['config']['per_gpu_batch'] == 5
    assert 'REHEARSAL' in rec['config']['parallelism'] and rec['vs_baseline'] is None
    assert abs(rec['value'] - 10 * 3 / (rec['ms_per_step'] * 3e-3)) / rec['value'] < 0.02      # whole-job images / MAX time
    # what the backend saw (a future SCALE line must prove N ranks on N distinct devices): the process group's own world
    # size, every rank's identity record and rows, its own chain and all-gather times
    rk = rec['ranks']
    assert rk['process_group'] == {'backend': 'gloo', 'world_size': 2}
    assert [r['rank'] for r in rk['per_rank']] == [0, 1] and [r['rows'] for r in rk['per_rank']] == [[0, 5], [5, 10]]
    assert all('host' in r and 'device' in r for r in rk['per_rank'])
    assert all(set(r['chain_ms']) == {'min', 'median', 'max'} and set(r['all_gather_ms']) == {'min', 'median', 'max'}
               for r in rk['per_rank'])
    assert rk['chain_ms']['max'] >= rk['chain_ms']['min'] >= 0 and rk['all_gather_ms']['max'] > 0
    assert rk['distinct_devices'] == 1            # both stub ranks report the host's 'cpu': ranks sharing a device are SEEN
    assert rec['config']['unet_forwards_per_sampler_step'] == 1 and rec['config']['images_through_the_unet_per_sampler_step'] == 5


@pytest.mark.parametrize('workload,batch,gbatch,cfgname', [('config2', 64, 512, 'configs[2]'), ('config5', 16, 128, 'configs[4]')])
def test_bench_eight_rank_branch_over_gloo(workload, batch, gbatch, cfgname):
    """The driver's SCALE run shape rehearsed on the CPU: bench.py --gpus 8 with WORLD_SIZE=8 over gloo, the denoiser stubbed out
    (ND_BENCH_STUB=1: no GPU work) -- BASELINE configs[2] (global batch 512 = 64 per rank) and configs[4] (128 = 16 per rank):
    rows per rank, gather order, the process group's own world size, whole-job images over the MAX-reduced time."""
    import json
    import subprocess
    port = _free_port()
    procs = []
    for r in range(8):
        env = dict(os.environ, WORLD_SIZE='8', RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), ND_BENCH_BACKEND='gloo', ND_BENCH_STUB='1', OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2',
                                       '--warmup', '1', '--workload', workload], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-400:] for o in outs]
    assert not any(l.startswith('{') for o in outs[1:] for l in o[0].splitlines()), 'only rank 0 prints the JSON line'
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['steps'] == 2 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == gbatch and rec['config']['per_gpu_batch'] == batch
    assert abs(rec['value'] - gbatch * 2 / (rec['ms_per_step'] * 2e-3)) / rec['value'] < 0.02
    rk = rec['ranks']
    assert rk['process_group'] == {'backend': 'gloo', 'world_size': 8}
    assert [r['rank'] for r in rk['per_rank']] == list(range(8))
    assert [r['rows'] for r in rk['per_rank']] == [[batch * r, batch * (r + 1)] for r in range(8)]


def _tune_worker(rank, world, port, tmp, q):
    sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
    import json
    import torch.distributed as dist
    from nicediffusion import _engine
    from nicediffusion.parallel import share_tuned_choices
    os.environ['ND_TUNE_CACHE'] = os.path.join(tmp, 'tune.json')
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
    try:
        if rank == 0:       # what rank 0 measured on ITS device (ordinal 0)
            _engine._TUNED[(0, 64, 64, 64, 192, 192, 3, 0, True, False)] = ('wino', 12)
            _engine._TUNED[(0, 'bf16', 16, 8, 8, 1024, 1024, 3, 0, False, True, False, 'sk1', 'es1', 'gnnb')] = ('bf16+splitk', 19, 4)
        else:               # a different choice this rank already holds (earlier local tuning / cache file) LOSES to rank 0's
            _engine._TUNED[(5, 64, 64, 64, 192, 192, 3, 0, True, False)] = ('wino', 8)
        taken = share_tuned_choices(device_index=5 if rank else 0)
        _engine._save_tune_cache()          # only rank 0 may write the shared file
        dist.barrier()
        q.put((rank, list(taken), sorted((list(map(str, k)), list(v)) for k, v in _engine._TUNED.items()),
               json.load(open(os.environ['ND_TUNE_CACHE']))))
    finally:
        dist.destroy_process_group()


def test_rank0_tuning_is_shared_and_only_rank0_writes_the_cache(tmp_path):
    """parallel.share_tuned_choices: rank 0's measured kernel choices reach the other rank re-keyed to its device ordinal
    (identical kernels on every rank => a sharded run is comparable bit for bit with the single-process one), a different
    entry the rank already holds is REPLACED by rank 0's (and counted), and ND_TUNE_CACHE is written by rank 0 alone."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, taken, tuned, cache = q.get(timeout=300)
        got[r] = (taken, tuned, cache)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == [0, 0] and got[1][0] == [2, 1]  # rank 1 took both choices; one replaced a different local one
    keys1 = {tuple(k): v for k, v in got[1][1]}
    assert keys1[tuple(map(str, (5, 64, 64, 64, 192, 192, 3, 0, True, False)))] == ['wino', 12]
    assert keys1[tuple(map(str, (5, 'bf16', 16, 8, 8, 1024, 1024, 3, 0, False, True, False, 'sk1', 'es1', 'gnnb')))] == ['bf16+splitk', 19, 4]
    assert got[0][2] == got[1][2] and len(got[0][2]) == 3       # the file holds rank 0's two entries + the library stamp, whoever reads it
    assert '__stamp__' in got[0][2]
