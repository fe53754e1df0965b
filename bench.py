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
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (same guide; the 5 PF figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0

# BASELINE.json configs.  The default (and the line the driver records) is configs[1]; --workload config4 / config5 run
# the bf16 configurations at their per-GPU batch on one GPU (their lines are kept under profiles/).
WORKLOADS = {
    # BASELINE configs[0] (the reference's own CPU-runnable case, plumbing only): EMNIST preset model, 50-step DDIM, batch 4.
    # The CPU side of this line is the oracle run IN FULL (3 repeats, median), not an extrapolation.
    'config1': dict(preset='EMNIST', chain=50, ddim=True, sched='cosine', batch=4, dtype='fp32', cfg=None, classes=27,
                    metric='sampled images/sec (50-step DDIM, EMNIST 28x28 UNet, batch 4)',
                    name='EMNIST 28x28 UNet (EMNIST preset, 18M params), {}-step DDIM eta=0, cosine schedule, '
                         'learned_interpolation, batch 4'),
    'config2': dict(preset='OPENAI_64', chain=250, ddim=True, sched='cosine', batch=64, dtype='fp32', cfg=None,
                    metric='sampled images/sec (250-step DDIM, 64x64 cond ImageNet UNet)',
                    name='64x64 conditional ImageNet UNet (OPENAI_64 preset, 296M params), {}-step DDIM eta=0, cosine '
                         'schedule, learned_interpolation'),
    'config4': dict(preset='OPENAI_128', chain=1000, ddim=False, sched='linear', batch=16, dtype='bf16', cfg=0.8,
                    metric='sampled images/sec (1000-step DDPM, classifier-free guidance, 128x128 cond ImageNet UNet, bf16)',
                    name='128x128 conditional ImageNet UNet (OPENAI_128 preset + null class, 422M params), {}-step DDPM '
                         'p_sample, linear schedule, learned_interpolation, classifier-free guidance w=0.8 (2B forwards '
                         'per step)'),
    'config5': dict(preset='OPENAI_256', chain=50, ddim=True, sched='linear', batch=16, dtype='bf16', cfg=None,
                    metric='sampled images/sec (50-step DDIM, 256x256 UNet, bf16)',
                    name='256x256 UNet (OPENAI_256 preset, random-init, 554M params), {}-step DDIM eta=0, linear schedule, '
                         'learned_interpolation'),
}
PRESET_STEPS = 250
PER_GPU_BATCH = 64


def synthetic_weights(model, seed=1234):
    """SURVEY 8(d)'s synthetic weights: numpy.random.default_rng(1234) normals in state_dict key order, sigma 0.02 for
    ordinary tensors, 0.005 for the tensors the reference zero-initialises (model.py:177,253,448), GroupNorm weight
    1 + 0.02 n -- the same tensors tests/ build with the oracle's synth_state_dict(seed=1234), so the full-size parity tests
    run the very model this file times.  (Throughput does not depend on the values, but all-zero outputs would let the chip
    clock higher than real data does.)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in model.state_dict().items():
            n = torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32))
            if name.endswith('norm.weight') or name == 'out.0.weight':
                p.copy_(1 + 0.02 * n)
            elif '.out_conv.' in name or '.proj_out.' in name or name.startswith('out.2.'):
                p.copy_(0.005 * n)
            else:
                p.copy_(0.02 * n)


def build(device, wl=None):
    from nicediffusion import default_args as DA
    from nicediffusion.model import DiffusionModel
    from nicediffusion.diffusion import Diffusion
    wl = WORKLOADS['config2'] if wl is None else wl
    margs = dict(getattr(DA, wl['preset'] + '_MODEL_ARGS'))
    if wl['cfg'] is not None:
        margs['num_classes'] += 1            # classifier-free guidance adds the null class (utils.py:211-212)
    model = DiffusionModel(**margs)
    synthetic_weights(model)
    model.to(device).eval()
    model.compute_dtype = wl['dtype']
    diff = Diffusion(model, original_num_steps=1000, rescaled_num_steps=wl['chain'], sampling_var_type='learned_interpolation',
                     loss_type='hybrid', beta_schedule=wl['sched'], use_ddim=wl['ddim'], ddim_eta=0.0 if wl['ddim'] else None,
                     guidance_method=None if wl['cfg'] is None else 'classifier_free', guidance_strength=wl['cfg'],
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


def _traffic_table():
    """Per-shape HBM traffic of the conv kernels from the PMC passes (tools/pmc_shapes.py -> profiles/rNN_pmc_shapes.json:
    one entry per (kernel kind, variant, ksize, NI, H, W, Cin, N) with FETCH_SIZE x2 + WRITE_SIZE per launch); rocprofv3 cannot run
    inside this process, so the table is regenerated by that script and looked up by the shapes actually launched."""
    for name in ('r06_pmc_shapes.json', 'r05_pmc_shapes.json', 'r04_pmc_shapes.json', 'r03_pmc_shapes.json', 'r02_pmc_shapes.json'):
        try:
            return json.load(open(os.path.join(ROOT, 'profiles', name))), name
        except (OSError, ValueError):
            continue
    return None, None


CONV_FNS = ('nd_conv_nhwc', 'nd_conv3x3_winograd_nhwc', 'nd_conv_bf16_nhwc', 'nd_conv3x3_winograd_stats_nhwc',
            'nd_conv3x3_winograd_vstats_nhwc', 'nd_conv3x3_bf16_stats_nhwc', 'nd_conv1x1_bf16_stats_nhwc', 'nd_conv_bf16_splitk_nhwc',
            'nd_conv_splitk_nhwc', 'nd_conv3x3_winograd_splitk_nhwc', 'nd_conv1x1_stats_nhwc', 'nd_conv3x3_winograd_f4_nhwc')
# flops EXECUTED on the matrix pipe / direct-convolution flops: Winograd F(2x2,3x3) runs 16 positions per 4 output pixels
# (4/9), F(4x4,3x3) 36 per 16 (1/4)
EXEC_FRACTION = {'wino': 4.0 / 9.0, 'wf4': 1.0 / 4.0}


def roofline_from(rows, lib, dtype='fp32', esize=4):
    """Dominant kernel = the conv kernel instantiation with the largest total time in one forward.  `achieved` / `frac`
    are EXECUTED matrix-pipe flops per second (the Winograd kernels execute 4/9 of the direct-convolution flops);
    the algorithmic (direct-convolution) rate is kept under `algorithmic_equivalent`."""
    import ctypes
    conv_fns = CONV_FNS
    peak = PEAK_BF16_TFLOPS if dtype == 'bf16' else PEAK_F32_TFLOPS

    def executed(r):
        return r['flops'] * (EXEC_FRACTION.get(r['variant'][0], 1.0) if r['variant'] else 1.0)
    groups = {}
    for r in rows:
        if r['fn'] not in conv_fns or not r.get('variant'):
            continue
        key = (tuple(r['variant']), r['ksize'])
        g = groups.setdefault(key, dict(ms=0.0, flops=0, exec=0.0, launches=0, rows=[]))
        g['ms'] += r['ms']
        g['flops'] += r['flops']
        g['exec'] += executed(r)
        g['launches'] += 1
        g['rows'].append(r)
    key, g = max(groups.items(), key=lambda kv: kv[1]['ms'])
    (kind, var), ksize = key
    bm, bn, nt, nsub, apf = (ctypes.c_int() for _ in range(5))
    if kind == 'wino':
        lib.nd_conv_winograd_variant_info(var, *(ctypes.byref(v) for v in (bm, bn, nt, nsub, apf)))
        kname = '{} (Winograd F(2x2,3x3) on fp32 MFMA; {} px x {} ch per block, {} threads)'.format(
            lib.nd_conv_winograd_variant_name(var).decode(), bm.value, bn.value, nt.value)
    elif kind == 'wf4':
        lib.nd_conv_winograd_f4_variant_info(var, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nt))
        kname = '{} (Winograd F(4x4,3x3) on fp32 MFMA v_mfma_f32_16x16x4_f32; {} px x {} ch per workgroup, {} threads)'.format(
            lib.nd_conv_winograd_f4_variant_name(var).decode(), bm.value, bn.value, nt.value)
    elif kind == 'bf16':
        lib.nd_conv_bf16_variant_info(max(var, 0), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nt))
        kname = '{}<{}x{} tile, {} threads, {} taps> ({})'.format(
            lib.nd_conv_bf16_variant_name(max(var, 0)).decode(), bm.value, bn.value, nt.value, ksize * ksize,
            'v_mfma_f32_16x16x32_bf16' if lib.nd_conv_bf16_variant_layout(max(var, 0)) else 'v_mfma_f32_32x32x16_bf16')
    else:
        lib.nd_conv_variant_info(var, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nt))
        base = 'nd::conv_mfma_kernel' if var < 9 else ('nd::gemm_stream_kernel' if var < 13 else ('nd::gemm_f32_kernel' if var == 13 else 'nd::gemm4_kernel'))
        kname = '{}<{}x{} tile, {} threads, {} taps>'.format(base, bm.value, bn.value, nt.value, ksize * ksize)
    # HBM traffic of this kernel over the forward: counter bytes of every shape it ran on / algorithmic bytes
    traffic = None
    tab, tab_name = _traffic_table()
    if tab:
        hb = ab = 0.0
        missing = 0
        for r in g['rows']:
            NI, H, W, Cin, N = r['shape']
            rec = tab.get('{}{}:{}:k{}:{}:{}:{}:{}:{}'.format(kind, '+stats' if 'stats' in r['fn'] else '', var, ksize, NI, H, W, Cin, N))
            alg = esize * (NI * H * W * (Cin + N) + ksize * ksize * Cin * N)
            if rec is None:
                missing += 1
                continue
            hb += rec['hbm_bytes']
            ab += alg
        if ab > 0:
            from nicediffusion import _engine
            stamps = tab.get('_stamps') or {}
            traffic = {'hbm_bytes_per_launch': hb / max(1, g['launches'] - missing), 'algorithmic_bytes_per_launch':
                       ab / max(1, g['launches'] - missing), 'ratio': round(hb / ab, 3), 'shapes_missing': missing,
                       # the table is a lookup, not a measurement of this run: stale = it was taken on another library build
                       'stale': _engine._tune_stamp() not in stamps.values(),
                       'source': 'profiles/{} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch, keyed by the shapes '
                                 'launched)'.format(tab_name)}
    sec = g['ms'] * 1e-3
    achieved = g['exec'] / sec / 1e12
    total_ms = sum(r['ms'] for r in rows)
    conv_ms = sum(v['ms'] for v in groups.values())
    conv_fl = sum(v['flops'] for v in groups.values())
    conv_ex = sum(v['exec'] for v in groups.values())
    return {
        'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4), 'traffic': traffic,
        'kernel': kname,
        'note': 'achieved = flops EXECUTED on the matrix pipe / time (Winograd F(2x2,3x3): 4/9 of the direct-convolution '
                'flops, F(4x4,3x3): 1/4); algorithmic_equivalent = direct-convolution flops / time',
        'algorithmic_equivalent': round(g['flops'] / sec / 1e12, 2),
        'launches_per_forward': g['launches'], 'avg_launch_ms': round(g['ms'] / g['launches'], 4),
        'flops_per_launch_avg': g['flops'] / g['launches'],
        'share_of_forward_time': round(g['ms'] / total_ms, 4),
        'all_mfma_conv': {'executed_tflops': round(conv_ex / (conv_ms * 1e-3) / 1e12, 2),
                          'frac_of_peak': round(conv_ex / (conv_ms * 1e-3) / 1e12 / peak, 4),
                          'algorithmic_tflops': round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                          'share_of_forward_time': round(conv_ms / total_ms, 4)},
    }, total_ms, conv_ex


def class_breakdown(rows):
    out = {}
    for r in rows:
        k = r['label'].split('.')[0] if r['fn'] not in CONV_FNS else r['label']
        if r['fn'].startswith('nd_groupnorm'):
            k = 'groupnorm_' + r['fn'].split('_')[2]
            if r['fn'] == 'nd_groupnorm_channel_partials_nhwc':
                k = 'groupnorm_stats'            # statistics = per-tensor channel partials + the folds into groups
        out[k] = out.get(k, 0.0) + r['ms']
    return {k: round(v, 3) for k, v in sorted(out.items(), key=lambda kv: -kv[1])}


def pick_cpu_threads(fn):
    """PyTorch's default thread count (the box's physical cores) is not the fastest for these small convolutions on a
    two-socket host: run ``fn`` (one small sampler step; also the warm-up of the thread pool and allocator) at the default
    and at 1/2, 1/4, 1/8, 1/16 of it (never below 8), leave the fastest count set and return (it, the default, {count:
    seconds}).  The fastest count uses a small part of a 256-CPU host: the line says how much (``host_cpus_used_frac``)."""
    default_threads = torch.get_num_threads()
    tried = {}
    fn()
    for th in sorted({default_threads} | {max(8, default_threads // d) for d in (2, 4, 8, 16)}, reverse=True):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        fn()
        tried[th] = round(time.perf_counter() - t0, 2)
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    return best, default_threads, tried


def cpu_baseline(model, margs, wl, batch=None, steps=2):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference, pinned against it in tests/) on this box's
    host cores.  configs[0] runs IN FULL (the whole 50-step chain at batch 4, 3 repeats, median).  The other workloads
    take a bounded sample (BASELINE.md section 4 / SURVEY 8(d)) -- 2 timed sampler steps (UNet forward(s) + update) at batch
    16 / 4 / 2 for the 64 / 128 / 256-pixel presets -- extrapolated to the whole chain.  Both run at the fastest of four
    thread counts (pick_cpu_threads); it is a reported baseline, not a target."""
    from oracle import unet_oracle as UO, diffusion_oracle as DO
    R = margs['resolution']
    ncls = wl.get('classes', 1000)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    so = DO.SamplerOracle(lambda a, b, c: UO.unet_forward(sd, margs, a, b, c), DO.Schedule(1000, wl['chain'], wl['sched']),
                          'learned_interpolation', use_ddim=wl['ddim'], ddim_eta=0.0 if wl['ddim'] else None,
                          guidance_method=None if wl['cfg'] is None else 'classifier_free', guidance_strength=wl['cfg'])
    if wl.get('preset') == 'EMNIST':
        B = wl['batch']
        torch.manual_seed(0)
        x = torch.randn(B, margs.get('in_channels', 1), R, R)
        y = (torch.arange(B) * 37) % ncls
        best_threads, default_threads, tried = pick_cpu_threads(lambda: so.ddim_step(x, wl['chain'] - 1, y))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            so.denoise(x, y)
            ts.append(time.perf_counter() - t0)
        torch.set_num_threads(default_threads)
        med = sorted(ts)[1]
        return {'value': round(B / med, 4), 'unit': 'images/sec', 'cores': best_threads, 'kind': 'port',
                'host_cpus': os.cpu_count(), 'runs_s': [round(t, 3) for t in ts], 'threads_tried_s_per_step': tried,
                'sample': 'the whole workload: {}-step DDIM chain at batch {} on {} threads of the host (the fastest of {}), 3 '
                          'repeats, median {:.2f} s'.format(wl['chain'], B, best_threads, sorted(tried), med)}
    if batch is None:
        batch = 16 if R <= 64 else (4 if R <= 128 else 2)
    torch.manual_seed(0)
    x = torch.randn(batch, 3, R, R)
    y = (torch.arange(batch) * 37) % ncls + (1 if wl['cfg'] is not None else 0)
    step = so.ddim_step if wl['ddim'] else so.ddpm_step
    t = wl['chain'] - 1
    nb = min(4, batch)
    best_threads, default_threads, tried = pick_cpu_threads(lambda: step(x[:nb], t, y[:nb]))
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        x, _ = step(x, t - i, y)
        ts.append(time.perf_counter() - t0)
    dt = sum(ts) / steps
    torch.set_num_threads(default_threads)
    nfwd = batch * (2 if wl['cfg'] is not None else 1)
    return {'value': round(batch / (dt * wl['chain']), 6), 'unit': 'images/sec', 'cores': best_threads,
            'kind': 'port', 'host_cpus': os.cpu_count(), 'host_cpus_used_frac': round(best_threads / max(1, os.cpu_count() or 1), 3),
            's_per_image_forward': round(dt / nfwd, 4),
            'step_s': [round(v, 2) for v in ts], 'threads_tried_s_per_batch{}_step'.format(nb): tried,
            'sample': '{} timed sampler step{} (UNet forward{} + update; small untimed steps before, which also pick the thread '
                      'count) at batch {} of the same {}x{} '
                      'preset on {} threads of the host, {:.2f} s/step = {:.3f} s per image-forward, extrapolated x{} steps'.format(
                          steps, '' if steps == 1 else 's', 's (2 per step, CFG)' if wl['cfg'] is not None else '', batch, R, R,
                          best_threads, dt, dt / nfwd, wl['chain'])}


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup's CPU quota where one is set (a 1-GPU
    slice of a 256-CPU host may be given far fewer)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_worker(args):
    """One of the concurrent oracle processes of ``cpu_baseline_aggregate`` (python bench.py --cpu-worker DIR ...): load the
    weights the parent saved, warm up, wait until every worker is ready, time the sampler steps, write the times."""
    from oracle import unet_oracle as UO, diffusion_oracle as DO
    wl = dict(WORKLOADS[args.workload])
    d, idx, nproc, threads, batch, steps = args.cpu_worker, args.worker_index, args.worker_count, args.worker_threads, args.batch, args.steps
    torch.set_num_threads(threads)
    blob = torch.load(os.path.join(d, 'weights.pt'))
    sd, margs = blob['sd'], blob['margs']
    R = margs['resolution']
    so = DO.SamplerOracle(lambda a, b, c: UO.unet_forward(sd, margs, a, b, c), DO.Schedule(1000, wl['chain'], wl['sched']),
                          'learned_interpolation', use_ddim=wl['ddim'], ddim_eta=0.0 if wl['ddim'] else None,
                          guidance_method=None if wl['cfg'] is None else 'classifier_free', guidance_strength=wl['cfg'])
    torch.manual_seed(idx)
    x = torch.randn(batch, 3, R, R)
    y = (torch.arange(batch) * 37) % wl.get('classes', 1000) + (1 if wl['cfg'] is not None else 0)
    step = so.ddim_step if wl['ddim'] else so.ddpm_step
    t = wl['chain'] - 1
    step(x[:min(4, batch)], t, y[:min(4, batch)])
    open(os.path.join(d, 'ready.{}'.format(idx)), 'w').close()
    deadline = time.time() + 300
    while time.time() < deadline and sum(os.path.exists(os.path.join(d, 'ready.{}'.format(i))) for i in range(nproc)) < nproc:
        time.sleep(0.05)
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        x, _ = step(x, t - i, y)
        ts.append(time.perf_counter() - t0)
    json.dump({'start': time.time() - sum(ts), 'step_s': ts}, open(os.path.join(d, 'done.{}.json'.format(idx)), 'w'))


def cpu_baseline_aggregate(model, margs, wl, threads, batch, steps=2):
    """"The box's best" on the host side is not one oracle process: N = usable CPUs // threads processes of ``threads``
    threads each, started together on the same bounded sample; aggregate = N * batch images per (slowest worker's mean
    step time x chain).  Reported beside the single-process figure (None when the slice has room for one process only)."""
    import shutil
    import subprocess
    import tempfile
    ncpu = usable_cpus()
    nproc = min(ncpu // max(threads, 1), 16)
    if nproc < 2:
        return {'processes': 1, 'usable_cpus': ncpu, 'note': 'the CPUs this job may use hold one {}-thread process: the '
                'aggregate is the single-process figure'.format(threads)}
    base = '/dev/shm' if os.path.isdir('/dev/shm') else None
    d = tempfile.mkdtemp(prefix='nd_cpu_', dir=base)
    try:
        torch.save({'sd': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'margs': margs}, os.path.join(d, 'weights.pt'))
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', d, '--worker-count', str(nproc), '--worker-threads',
               str(threads), '--batch', str(batch), '--steps', str(steps), '--workload', wl['key']]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        procs = [subprocess.Popen(cmd + ['--worker-index', str(i)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                 for i in range(nproc)]
        t_end = time.time() + 240
        for pr in procs:
            try:
                pr.wait(timeout=max(1.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()
        res = []
        for i in range(nproc):
            try:
                res.append(json.load(open(os.path.join(d, 'done.{}.json'.format(i)))))
            except (OSError, ValueError):
                pass
        if len(res) < nproc:
            err = b''.join(pr.stderr.read()[-300:] for pr in procs if pr.returncode not in (0, None))
            return {'processes': nproc, 'usable_cpus': ncpu, 'failed': nproc - len(res), 'stderr_tail': err.decode('utf-8', 'replace')[-300:]}
        worst = max(sum(r['step_s']) / len(r['step_s']) for r in res)
        return {'value': round(nproc * batch / (worst * wl['chain']), 6), 'unit': 'images/sec', 'processes': nproc,
                'threads_per_process': threads, 'cores': nproc * threads, 'usable_cpus': ncpu,
                'step_s_by_worker': [round(sum(r['step_s']) / len(r['step_s']), 2) for r in res],
                'start_spread_s': round(max(r['start'] for r in res) - min(r['start'] for r in res), 2),
                'sample': '{} concurrent oracle processes x {} threads, {} timed batch-{} sampler steps each after a common '
                          'start; slowest worker {:.2f} s/step, extrapolated x{} steps'.format(nproc, threads, steps, batch, worst, wl['chain'])}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# measured on MI355X against the REAL reference's goldens (tests/test_gpu_model.py): what `dtype: f32`
# means on the headline line since the 3x3 layers run Winograd F(4x4,3x3), the numerically loosest fp32 arithmetic here
FP32_PRECISION = {
    'arithmetic': 'fp32 end to end (v_mfma_f32_* = exact fp32 FMA chains); Winograd F(4x4,3x3) on the 3x3 layers the tune cache '
                  'gives it (72 of 74 at B = 64), F(2x2,3x3) / direct elsewhere',
    'forward_max_abs_vs_reference': {'measured': '5.6e-6..7.5e-6 (B = 64 plan, rows 0/31/63; output absmax 0.63)', 'test_bound': 1e-4},
    'teacher_forced_ddim_step_max_abs': {'measured': '6.7e-8..1.8e-5 (indices 249/125/1/0 of the 250-step chain)', 'test_bound': 1e-4},
    'free_running_25_step_preset_chain_max_abs': {'measured': '1.12e-4 (F(4x4) forced on 72 of 74 layers, B = 2)', 'test_bound': 2.3e-4},
    'tolerance_required': '1e-3 (BASELINE.json north_star)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1, help='timed passes (each = one whole sampling chain per GPU)')
    ap.add_argument('--warmup', type=int, default=1, help='untimed passes')
    ap.add_argument('--workload', default='config2', choices=sorted(WORKLOADS),
                    help='BASELINE.json configuration: config2 = configs[1] (the headline metric, default); config4 / config5 = '
                         'configs[3] / [4], bf16, at their per-GPU batch')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the workload\'s)')
    ap.add_argument('--chain', type=int, default=None, help='(debug) sampler steps per pass; the metric needs the full chain')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='(debug/profiling) launch every step eagerly instead of hipGraph replay')
    ap.add_argument('--retune', action='store_true', help='ignore profiles/tune_cache_<workload>.json and measure the tile variants here')
    ap.add_argument('--save-tune-cache', default=None, metavar='FILE', help='write the kernel choices this run used (rank 0)')
    ap.add_argument('--cpu-worker', default=None, metavar='DIR', help=argparse.SUPPRESS)      # cpu_baseline_aggregate's child
    ap.add_argument('--worker-index', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--worker-count', type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument('--worker-threads', type=int, default=16, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)
    wl = dict(WORKLOADS[args.workload], key=args.workload)
    full_chain = wl['chain']
    if args.chain is None:
        args.chain = full_chain
    if not 0 < args.chain <= full_chain:
        raise SystemExit('--chain must be in 1..{} for {}'.format(full_chain, args.workload))
    if args.batch is None:
        args.batch = wl['batch']

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
    # a process group whenever a launcher set the rendezvous up (torchrun exports RANK / MASTER_PORT even for one rank), so a
    # one-GPU box can run this file's N > 1 code -- RCCL init, the tuning broadcast, the all-gather, the MAX all-reduce --
    # with a world of one; the plain `python bench.py` of the N = 1 measurement has neither variable and no process group
    if world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ):
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
        margs, model, diff = dict(resolution=64), None, _StubDiffusion()
    else:
        margs, model, diff = build(device, wl)
        diff.seed = 1234                       # DDPM: in-kernel Philox noise, same seed on every rank (rows keyed globally)
    if args.no_graph:
        diff.use_graph = False
    B = args.batch
    Bg = B * world
    R = margs['resolution']
    # global batch generated identically on every rank, then sliced (an N-GPU run is comparable row by row)
    torch.manual_seed(0)
    x_global = torch.randn(Bg, margs.get('in_channels', 3), R, R)
    y_global = (torch.arange(Bg) * 37) % wl.get('classes', 1000) + (1 if wl['cfg'] is not None else 0)   # CFG: label 0 is the null class
    from nicediffusion.parallel import shard_slice, all_gather_rows
    sl = shard_slice(Bg, rank, world)
    x_local = x_global[sl].to(device)
    y_local = y_global[sl].to(device)
    if not stub:
        diff.first_row = sl.start

    rank_marks = []            # per timed pass: (chain start, chain end = gather start, gather end) on the launch stream

    def mark():
        if stub:
            return time.perf_counter()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def one_pass(timed=False):
        m0 = mark() if timed else None
        out = diff.denoise(x=x_local, kwargs={'y': y_local}, batch_size=B, steps_to_do=args.chain, progress=False)
        m1 = mark() if timed else None
        if dist is not None:
            out = all_gather_rows(out, Bg, rank, world)
        if timed:
            rank_marks.append((m0, m1, mark()))
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    tune_cache = None
    if not stub:
        # the committed choices of this workload's plan (profiles/tune_cache_<workload>.json, written by --save-tune-cache on
        # an MI355X): the run starts without ~20 s of tuning launches and every box runs the SAME kernels per layer; a file
        # stamped by another library build is ignored and the plan is tuned here.  ND_TUNE_CACHE / --retune bypass it
        from nicediffusion import _engine
        path = os.path.join(ROOT, 'profiles', 'tune_cache_{}.json'.format(args.workload))
        if not args.retune and not os.environ.get('ND_TUNE_CACHE') and args.batch == wl['batch']:
            n = _engine.preload_tune_cache(path, device_index=device.index)
            tune_cache = {'file': os.path.relpath(path, ROOT), 'choices_loaded': n}
    if dist is not None and not stub:
        # rank 0 measures the tile variants, every rank runs its choices (identical kernels on every GPU)
        from nicediffusion.parallel import tune_on_rank0
        tune_on_rank0(model, (2 if wl['cfg'] is not None else 1) * B)
    def progress(what, i, n):
        # a line per pass on stderr (rank 0): long runs (the driver's 25 chains of ~17 s) stay visibly alive; no device
        # synchronisation -- the line marks that the pass has been ENQUEUED
        if rank == 0:
            print('bench: {} pass {}/{} enqueued at {:.1f} s'.format(what, i + 1, n, time.perf_counter() - t_start),
                  file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    for i in range(args.warmup):
        one_pass()
        progress('warm-up', i, args.warmup)
    barrier()
    # per-pass marks for the median: events on the launch stream, no synchronisation inside the timed region
    marks = [] if stub else [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    if marks:
        marks[0].record()
    for i in range(args.steps):
        out = one_pass(timed=True)
        if marks:
            marks[i + 1].record()
        progress('timed', i, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    pass_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)] if marks else []
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all()
    # what the backend saw: every rank's device identity and its own chain / all-gather times, gathered to rank 0
    from nicediffusion.parallel import device_identity
    if stub:
        span = lambda a, b: (b - a) * 1e3
    else:
        span = lambda a, b: a.elapsed_time(b)
    me = dict(device_identity(device), rank=rank, local_rank=local_rank, rows=[sl.start, sl.stop],
              chain_ms=[round(span(a, b), 2) for a, b, _ in rank_marks],
              all_gather_ms=[round(span(b, c), 3) for _, b, c in rank_marks])
    ranks = [me]
    if dist is not None:
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
    if not stub and args.save_tune_cache and rank == 0:
        from nicediffusion import _engine
        _engine._save_tune_cache(args.save_tune_cache)
    if stub:
        assert out.shape[0] == Bg and torch.equal(out, x_global + 1), 'gathered rows are not in global order'
        args.no_breakdown = args.no_cpu_baseline = True

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = Bg * args.steps / dt * (args.chain / full_chain)   # == Bg*steps/dt for the real full-length chain
        fwd_per_step = 2 if wl['cfg'] is not None else 1
        line = {
            'metric': wl['metric'], 'value': round(value, 4),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if wl['dtype'] == 'fp32' else 'bf16',
            'data': 'synthetic (random-init weights, randn x_T, labels (arange*37)%{})'.format(wl.get('classes', 1000)),
            'config': {'workload': wl['name'].format(args.chain), 'baseline_config': args.workload,
                       'per_gpu_batch': B, 'global_batch': Bg, 'sampler_steps_per_pass': args.chain,
                       'unet_forwards_per_sampler_step': fwd_per_step,
                       'images_through_the_unet_per_sampler_step': fwd_per_step * B,
                       'parallelism': ('batch-shard x{} + all-gather'.format(world) if world > 1 else 'single GPU') +
                                      ('' if backend == 'nccl' else ' [REHEARSAL backend={}{}: not a measurement]'.format(
                                          backend, ', stub denoiser on CPU' if stub else '')),
                       'loop': 'hipGraph replay' if diff.use_graph else 'eager'},
            'ms_per_sampler_step': round(ms_per_step / args.chain, 3),
        }
        if pass_ms:
            srt = sorted(pass_ms)
            med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
            # `value` is the contract's mean over the timed region; the median of the passes (HIP events) is beside it
            line['passes'] = {'ms': [round(v, 1) for v in pass_ms], 'median_ms': round(med, 2),
                              'median_images_per_sec': round(Bg / (med * 1e-3) * (args.chain / full_chain), 4)}
        def stats(vals):
            v = sorted(vals)
            if not v:
                return None
            med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
            return {'min': round(v[0], 3), 'median': round(med, 3), 'max': round(v[-1], 3)}
        ident = lambda r: (r.get('host'), r.get('uuid'), r.get('pci_domain_id'), r.get('pci_bus_id'), r.get('pci_device_id'),
                           r.get('device'))
        line['ranks'] = {
            'process_group': None if dist is None else {'backend': dist.get_backend(), 'world_size': dist.get_world_size()},
            'distinct_devices': len({ident(r) for r in ranks}),
            'chain_ms': stats([v for r in ranks for v in r['chain_ms']]),
            'all_gather_ms': stats([v for r in ranks for v in r['all_gather_ms']]) if dist is not None else None,
            'per_rank': [{k: (stats(v) if k in ('chain_ms', 'all_gather_ms') else v) for k, v in r.items()} for r in ranks]}
        if tune_cache is not None:
            line['config']['tune_cache'] = tune_cache
        if not stub:
            from nicediffusion import _hip
            # which library build and which plan switches produced the number: the source hash every measured artefact is
            # stamped with, and every ND_* variable set in the environment (none = every switch at its default)
            line['build_id'] = _hip.build_id()
            line['config']['switches'] = {k: v for k, v in sorted(os.environ.items()) if k.startswith('ND_')}
            if tune_cache is not None:
                line['config']['tune_cache']['stale'] = tune_cache['choices_loaded'] == 0
        if wl['dtype'] == 'fp32' and args.workload == 'config2':
            line['config']['precision'] = FP32_PRECISION
        if wl['dtype'] == 'bf16':
            # no bf16 reference exists: the bounds are this build's own, each <= 2x the value measured on
            # MI355X against the reference's fp32 goldens (tests/test_gpu_bf16.py)
            line['config']['precision'] = {
                'storage': 'bf16 activations and weights in HBM', 'accumulation': 'fp32',
                'kept_fp32': 'GroupNorm statistics, embedding MLP, x_t / model output / sampler update',
                'forward_rel_rms_vs_fp32_reference': {'measured': '8.8e-3..1.5e-2 (presets)', 'test_bound': '1.76e-2..3.0e-2'},
                'teacher_forced_step_max_abs': {'measured': 6.1e-3, 'test_bound': 1.2e-2},
                'free_running_10_step_chain_max_abs': {'measured': '5.9e-3..3.7e-2', 'test_bound': '1.3e-2..7e-2'}}
        if not args.no_breakdown:
            NI = fwd_per_step * B
            plan, rows = kernel_breakdown(model, NI)
            roof, fwd_ms, conv_exec = roofline_from(rows, plan.lib, wl['dtype'], 2 if wl['dtype'] == 'bf16' else 4)
            line['roofline'] = roof
            peak = PEAK_BF16_TFLOPS if wl['dtype'] == 'bf16' else PEAK_F32_TFLOPS
            other_fl = plan.flops - sum(r['flops'] for r in rows if r.get('variant') and r['fn'] != 'nd_attention_nhwc')
            exec_fl = conv_exec + max(other_fl, 0)      # attention / linears execute their algorithmic flops
            line['forward'] = {'images': NI, 'eager_sum_of_kernels_ms': round(fwd_ms, 3),
                               'algorithmic_tflop': round(plan.flops / 1e12, 4),
                               'algorithmic_tflops': round(plan.flops / (fwd_ms * 1e-3) / 1e12, 2),
                               'executed_tflop': round(exec_fl / 1e12, 4),
                               'executed_frac_of_matrix_peak': round(exec_fl / (fwd_ms * 1e-3) / 1e12 / peak, 4),
                               'launches': len(rows), 'ms_by_class': class_breakdown(rows)}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(model, margs, wl)
            if wl.get('preset') != 'EMNIST':
                R_ = margs['resolution']
                line['cpu_baseline']['all_usable_cores'] = cpu_baseline_aggregate(
                    model, margs, wl, line['cpu_baseline']['cores'], 16 if R_ <= 64 else (4 if R_ <= 128 else 2))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()              # the other ranks wait here while rank 0 measures the per-kernel breakdown
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
