#!/usr/bin/env python3
"""Headline benchmark: sampled images/sec, 250-step DDIM, 64x64 class-conditional ImageNet UNet, fp32.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: a full 250-step DDIM chain (250 UNet forwards + sampler updates,
replayed as a hipGraph) for 64 images per GPU, x_T already resident in HBM.  With N > 1 every rank denoises its own
64 rows of the global batch (weak scaling; no communication inside the loop) and the finished samples are all-gathered
over RCCL inside the timed region.  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1] (configs[2] for N = 8); synthetic data: random-init weights of that architecture
(the reference's zero-initialised layers re-randomised so the output is not identically zero), x_T = randn, labels
(arange*37) % 1000.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'nice-diffusion_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL needs it on this driver

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X dense fp32 (vector = matrix) peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

PRESET_STEPS = 250
PER_GPU_BATCH = 64


def synthetic_weights(model, seed=1234):
    """Random weights of the architecture: N(0, 0.02) everywhere, N(0, 0.005) for the layers the reference
    zero-initialises, GroupNorm weight 1 + 0.02 n (throughput does not depend on the values, but all-zero outputs
    would let the chip clock higher than real data does)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            n = torch.randn(p.shape, generator=g)
            if name.endswith('norm.weight') or name == 'out.0.weight':
                p.copy_(1 + 0.02 * n)
            elif '.out_conv.' in name or '.proj_out.' in name or name.startswith('out.2.'):
                p.copy_(0.005 * n)
            else:
                p.copy_(0.02 * n)


def build(device, small=False):
    from nicediffusion.default_args import OPENAI_64_MODEL_ARGS
    from nicediffusion.model import DiffusionModel
    from nicediffusion.diffusion import Diffusion
    margs = dict(OPENAI_64_MODEL_ARGS)
    model = DiffusionModel(**margs)
    synthetic_weights(model)
    model.to(device).eval()
    diff = Diffusion(model, original_num_steps=1000, rescaled_num_steps=PRESET_STEPS, sampling_var_type='learned_interpolation',
                     loss_type='hybrid', beta_schedule='cosine', use_ddim=True, ddim_eta=0.0, guidance_method=None,
                     device=device)
    return margs, model, diff


def kernel_breakdown(model, batch, reps=2):
    """Per-launch HIP-event timing of one eager forward (events recorded on the launch stream)."""
    plan = model._plan(batch)
    plan.run()
    torch.cuda.synchronize()
    acc = None
    for _ in range(reps):
        rows = plan.run_timed()
        if acc is None:
            acc = rows
        else:
            for a, r in zip(acc, rows):
                a['ms'] += r['ms']
    for a in acc:
        a['ms'] /= reps
    return plan, acc


def roofline_from(rows, lib):
    """Dominant kernel = the conv_mfma_kernel instantiation with the largest total time in one forward."""
    import ctypes
    groups = {}
    for r in rows:
        if r['fn'] not in ('nd_conv_nhwc', 'nd_conv3x3_winograd_nhwc'):
            continue
        key = (r['variant'], r['ksize'])
        g = groups.setdefault(key, dict(ms=0.0, flops=0, launches=0))
        g['ms'] += r['ms']
        g['flops'] += r['flops']
        g['launches'] += 1
    key, g = max(groups.items(), key=lambda kv: kv[1]['ms'])
    (kind, var), ksize = key
    if kind == 'wino':
        bm, bn, nt, nsub, apf = (ctypes.c_int() for _ in range(5))
        lib.nd_conv_winograd_variant_info(var, *(ctypes.byref(v) for v in (bm, bn, nt, nsub, apf)))
        kname = '{} (Winograd F(2x2,3x3) on fp32 MFMA; {} px x {} ch per block, {} threads)'.format(
            lib.nd_conv_winograd_variant_name(var).decode(), bm.value, bn.value, nt.value)
        executed = 4.0 / 9.0
    else:
        bm, bn, nt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.nd_conv_variant_info(var, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nt))
        kname = 'nd::conv_mfma_kernel<{}x{} tile, {} threads, {} taps>'.format(bm.value, bn.value, nt.value, ksize * ksize)
        executed = 1.0
    traffic = None
    try:      # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')))
        for name, rec in pmc.items():
            if isinstance(rec, dict) and kname.startswith(name) and 'hbm_bytes' in rec:
                traffic = {'hbm_bytes_per_launch': rec['hbm_bytes'], 'algorithmic_bytes': rec['algorithmic_bytes'],
                           'shape': rec['shape'], 'source': 'profiles/r01_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE)'}
    except (OSError, ValueError):
        pass
    achieved = g['flops'] / (g['ms'] * 1e-3) / 1e12
    total_ms = sum(r['ms'] for r in rows)
    conv_ms = sum(v['ms'] for v in groups.values())
    conv_fl = sum(v['flops'] for v in groups.values())
    return {
        'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / PEAK_F32_TFLOPS, 4), 'traffic': traffic,
        'kernel': kname,
        'note': 'achieved = algorithmic (direct-convolution) flops / time; the Winograd kernel executes 4/9 of them '
                'on the matrix pipe, so frac can exceed 1' if kind == 'wino' else 'achieved = algorithmic flops / time',
        'mfma_pipe_frac': round(achieved * executed / PEAK_F32_TFLOPS, 4),
        'launches_per_forward': g['launches'], 'avg_launch_ms': round(g['ms'] / g['launches'], 4),
        'flops_per_launch_avg': g['flops'] / g['launches'],
        'share_of_forward_time': round(g['ms'] / total_ms, 4),
        'all_mfma_conv': {'achieved': round(conv_fl / (conv_ms * 1e-3) / 1e12, 2), 'share_of_forward_time':
                          round(conv_ms / total_ms, 4)},
    }, total_ms


def class_breakdown(rows):
    out = {}
    for r in rows:
        k = r['label'].split('.')[0] if r['fn'] not in ('nd_conv_nhwc', 'nd_conv3x3_winograd_nhwc') else r['label']
        if r['fn'].startswith('nd_groupnorm'):
            k = 'groupnorm_' + r['fn'].split('_')[2]
        out[k] = out.get(k, 0.0) + r['ms']
    return {k: round(v, 3) for k, v in sorted(out.items(), key=lambda kv: -kv[1])}


def cpu_baseline(model, diff, margs, batch=8, steps=2):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference, pinned against it in tests/) on this box's
    host cores: `steps` DDIM steps at B=`batch` of the same preset, extrapolated to the 250-step chain."""
    from oracle import unet_oracle as UO, diffusion_oracle as DO
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    so = DO.SamplerOracle(lambda a, b, c: UO.unet_forward(sd, margs, a, b, c), DO.Schedule(1000, PRESET_STEPS, 'cosine'),
                          'learned_interpolation', use_ddim=True, ddim_eta=0.0)
    torch.manual_seed(0)
    x = torch.randn(batch, 3, 64, 64)
    y = (torch.arange(batch) * 37) % 1000
    t = PRESET_STEPS - 1
    x, _ = so.ddim_step(x, t, y)          # warm-up (thread pool, oneDNN primitive cache)
    t0 = time.perf_counter()
    for i in range(steps):
        x, _ = so.ddim_step(x, t - 1 - i, y)
    dt = (time.perf_counter() - t0) / steps
    return {'value': round(batch / (dt * PRESET_STEPS), 6), 'unit': 'images/sec', 'cores': torch.get_num_threads(),
            'kind': 'port', 'host_cpus': os.cpu_count(),
            'sample': '{} DDIM steps (UNet forward + update) at batch {} of the same 64x64 preset on the host cores, '
                      '{:.2f} s/step, extrapolated x{} steps'.format(steps, batch, dt, PRESET_STEPS)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1, help='timed passes (each = one 250-step DDIM chain per GPU)')
    ap.add_argument('--warmup', type=int, default=1, help='untimed passes')
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='images per GPU')
    ap.add_argument('--chain', type=int, default=PRESET_STEPS, help='(debug) DDIM steps per pass; the metric needs 250')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='(debug/profiling) launch every step eagerly instead of hipGraph replay')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node {} for --gpus {}'.format(args.gpus, args.gpus))
    # ND_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the gather
    # goes through host memory); the measured configuration is always nccl = RCCL, one rank per GPU
    backend = os.environ.get('ND_BENCH_BACKEND', 'nccl')
    # ND_BENCH_STUB=1 (tests/test_distributed_gloo.py only): no GPU work at all -- the denoiser is replaced by
    # "x + 1" on host tensors so that THIS file's N > 1 control flow (sharding, gather, barrier, MAX-reduced time, the
    # JSON line) can be exercised with world_size 2 over gloo on a CPU-only machine.  It is never a measurement.
    stub = os.environ.get('ND_BENCH_STUB') == '1'
    if stub:
        assert backend == 'gloo', 'the stub rehearsal runs over gloo only'
        device = torch.device('cpu')
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        if backend != 'nccl':
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)      # RCCL on ROCm
        else:
            dist.init_process_group(backend)

    if stub:
        class _StubDiffusion:
            use_graph = False

            def denoise(self, x, kwargs, batch_size, steps_to_do, progress):
                assert x.shape[0] == batch_size == kwargs['y'].shape[0]
                return x + 1
        margs, model, diff = None, None, _StubDiffusion()
    else:
        margs, model, diff = build(device)
    if args.no_graph:
        diff.use_graph = False
    B = args.batch
    Bg = B * world
    # global batch generated identically on every rank, then sliced (an N-GPU run is comparable row by row)
    torch.manual_seed(0)
    x_global = torch.randn(Bg, 3, 64, 64)
    y_global = (torch.arange(Bg) * 37) % 1000
    from nicediffusion.parallel import shard_slice, all_gather_rows
    sl = shard_slice(Bg, rank, world)
    x_local = x_global[sl].to(device)
    y_local = y_global[sl].to(device)

    def one_pass():
        out = diff.denoise(x=x_local, kwargs={'y': y_local}, batch_size=B, steps_to_do=args.chain, progress=False)
        if world > 1:
            out = all_gather_rows(out, Bg, rank, world)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all()
    if stub:
        assert out.shape[0] == Bg and torch.equal(out, x_global + 1), 'gathered rows are not in global order'
        args.no_breakdown = args.no_cpu_baseline = True

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = Bg * args.steps / dt * (args.chain / PRESET_STEPS)   # == Bg*steps/dt for the real 250-step chain
        line = {
            'metric': 'sampled images/sec (250-step DDIM, 64x64 cond ImageNet UNet)', 'value': round(value, 4),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic (random-init weights, randn x_T, labels (arange*37)%1000)',
            'config': {'workload': '64x64 conditional ImageNet UNet (OPENAI_64 preset, 296M params), {}-step DDIM '
                                   'eta=0, cosine schedule, learned_interpolation'.format(args.chain),
                       'per_gpu_batch': B, 'global_batch': Bg, 'ddim_steps_per_pass': args.chain,
                       'parallelism': ('batch-shard x{} + all-gather'.format(world) if world > 1 else 'single GPU') +
                                      ('' if backend == 'nccl' else ' [REHEARSAL backend={}{}: not a measurement]'.format(
                                          backend, ', stub denoiser on CPU' if stub else '')),
                       'loop': 'hipGraph replay' if diff.use_graph else 'eager'},
            'ms_per_unet_forward_plus_update': round(ms_per_step / args.chain, 3),
        }
        if not args.no_breakdown:
            plan, rows = kernel_breakdown(model, B)
            roof, fwd_ms = roofline_from(rows, plan.lib)
            line['roofline'] = roof
            line['forward'] = {'eager_sum_of_kernels_ms': round(fwd_ms, 3), 'algorithmic_tflop': round(plan.flops / 1e12, 4),
                               'tflops': round(plan.flops / (fwd_ms * 1e-3) / 1e12, 2),
                               'frac_of_f32_peak': round(plan.flops / (fwd_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4),
                               'launches': len(rows), 'ms_by_class': class_breakdown(rows)}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(model, diff, margs)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()              # the other ranks wait here while rank 0 measures the per-kernel breakdown
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
