#!/usr/bin/env python3
"""Generate tests/golden/* by importing the REAL reference (runs only in the build container).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/gen_golden.py

The reference at /root/reference is put first on sys.path, so ``nicediffusion`` below is the reference's own
package (the product package lives under ``nice-diffusion_amd/`` and is NOT on the path here).  Nothing from
the reference is copied: only inputs (seeds, small tensors) and the outputs it computed are written.

Fixture design follows SURVEY.md 8(c): synthetic weights from numpy's default_rng in state_dict order with the
zero-initialised tensors overwritten, stored x_T and per-step noise, per-layer intermediates, teacher-forced
per-step vectors plus free-running loops on contractive weights.
"""
import json
import os
import sys

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nicediffusion.model import DiffusionModel  # noqa: E402  (reference)
from nicediffusion.diffusion import Diffusion  # noqa: E402  (reference)
from nicediffusion import default_args as ref_presets  # noqa: E402
from nicediffusion.utils import make_argparser, get_dicts_from_args  # noqa: E402

from oracle import unet_oracle as UO  # noqa: E402
from oracle import diffusion_oracle as DO  # noqa: E402
from tests.cases import TINY_CFGS, SCHEDULE_CASES, SAMPLER_CASES, CLI_CASES, labels_for  # noqa: E402

assert DiffusionModel.__module__ == 'nicediffusion.model' and \
    sys.modules['nicediffusion'].__path__[0].startswith(REF), 'reference not first on path'

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def ref_model(cfg, sd):
    m = DiffusionModel(**cfg)
    m.load_state_dict(sd, strict=True)        # also pins oracle.param_shapes key names
    return m.eval()


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: np.asarray(v) for k, v in arrs.items()})
    print('wrote', name, len(arrs), 'arrays')


# ------------------------------------------------------------------------------------------------ schedules (A1)
def gen_schedules():
    out = {}
    for name, (T, S, sched) in SCHEDULE_CASES.items():
        torch.manual_seed(0)
        m = DiffusionModel(resolution=8, in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                           attention_resolutions=(), channel_mult=(1,))
        d = Diffusion(m, T, S, 'small', 'simple', beta_schedule=sched, device=torch.device('cpu'))
        for attr in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
                     'sqrt_one_minus_alphas_cumprod', 'sqrt_reciprocal_alphas_cumprod',
                     'sqrt_reciprocal_alphas_minus_one_cumprod', 'posterior_mean_coef_x0', 'posterior_mean_coef_xt',
                     'posterior_variance', 'log_posterior_var_clipped'):
            out['{}/{}'.format(name, attr)] = getattr(d, attr)
        out['{}/timestep_map'.format(name)] = d.timestep_map.numpy()
    save('schedules.npz', **out)


# ------------------------------------------------------------------------------------------------ tiny forwards
def gen_tiny_forwards():
    for name, cfg in TINY_CFGS.items():
        sd = UO.synth_state_dict(cfg, seed=1234)
        m = ref_model(cfg, sd)
        B = 3
        g = np.random.default_rng(7)
        R = cfg['resolution']
        x = torch.from_numpy(g.standard_normal((B, cfg['in_channels'], R, R)).astype(np.float32))
        t = torch.tensor([2, 501, 998][:B], dtype=torch.long)
        y = labels_for(cfg, B)
        taps = {}
        hooks = []

        def mk(nm):
            def hook(mod, inp, outp):
                taps[nm] = outp.detach().clone()
            return hook
        for nm, mod in m.named_modules():
            # blocks only (downsampling.i.j / middle_block.j / upsampling.i.j) -> one golden per HIP op group
            parts = nm.split('.')
            if (parts[0] in ('downsampling', 'upsampling') and len(parts) == 3) or \
                    (parts[0] == 'middle_block' and len(parts) == 2):
                hooks.append(mod.register_forward_hook(mk(nm)))
        with torch.no_grad():
            out = m(x, t, y) if y is not None else m(x, t)
        for h in hooks:
            h.remove()
        # oracle agreement is asserted at generation time too
        o_taps = {}
        o = UO.unet_forward(sd, cfg, x, t, y, taps=o_taps)
        err = (o - out).abs().max().item()
        assert err < 2e-5, (name, err)
        arrs = dict(x=x.numpy(), t=t.numpy(), out=out.numpy())
        if y is not None:
            arrs['y'] = y.numpy()
        for k, v in taps.items():
            assert (o_taps[k] - v).abs().max().item() < 2e-5, (name, k)
            arrs['tap/' + k] = v.numpy()
        save('fwd_{}.npz'.format(name), **arrs)
        print('  ', name, 'oracle-vs-reference max err', err, 'out absmax', out.abs().max().item())


# ------------------------------------------------------------------------------------------------ presets
def gen_presets():
    meta = {}
    for pname, margs in (('EMNIST', ref_presets.EMNIST_MODEL_ARGS), ('OPENAI_64', ref_presets.OPENAI_64_MODEL_ARGS),
                         ('OPENAI_128', ref_presets.OPENAI_128_MODEL_ARGS),
                         ('OPENAI_256', ref_presets.OPENAI_256_MODEL_ARGS)):
        m = DiffusionModel(**margs)
        sdref = m.state_dict()
        shapes = UO.param_shapes(dict(margs))
        assert list(shapes.keys()) == list(sdref.keys()), pname
        assert all(tuple(sdref[k].shape) == tuple(v) for k, v in shapes.items()), pname
        meta[pname] = dict(n_tensors=len(sdref), n_params=int(sum(v.numel() for v in sdref.values())),
                           keys=list(sdref.keys()), shapes=[list(v.shape) for v in sdref.values()])
        del m
    with open(os.path.join(OUT, 'preset_state_dicts.json'), 'w') as f:
        json.dump(meta, f)
    print('wrote preset_state_dicts.json', {k: (v['n_tensors'], v['n_params']) for k, v in meta.items()})

    # EMNIST preset forward, B=2 (SURVEY 8(d) synthetic weights)
    cfg = dict(ref_presets.EMNIST_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    m = ref_model(cfg, sd)
    torch.manual_seed(0)
    x = torch.randn(2, 1, 28, 28)
    t = torch.tensor([10, 990])
    y = torch.tensor([3, 26])
    with torch.no_grad():
        out = m(x, t, y)
    o = UO.unet_forward(sd, cfg, x, t, y)
    assert (o - out).abs().max().item() < 2e-5
    save('fwd_preset_emnist.npz', x=x.numpy(), t=t.numpy(), y=y.numpy(), out=out.numpy())

    # 64x64 preset forward, B=1
    cfg = dict(ref_presets.OPENAI_64_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    m = ref_model(cfg, sd)
    torch.manual_seed(0)
    x = torch.randn(1, 3, 64, 64)
    t = torch.tensor([498])
    y = torch.tensor([37])
    with torch.no_grad():
        out = m(x, t, y)
    o = UO.unet_forward(sd, cfg, x, t, y)
    err = (o - out).abs().max().item()
    print('   64x64 preset oracle-vs-reference', err, 'absmax', out.abs().max().item())
    assert err < 5e-5
    save('fwd_preset_64.npz', x=x.numpy(), t=t.numpy(), y=y.numpy(), out=out.numpy())


# ------------------------------------------------------------------------------------------------ init parity
def gen_init():
    cfg = TINY_CFGS['adagn_updown']
    torch.manual_seed(0)
    m = DiffusionModel(**cfg)
    sd = m.state_dict()
    keys = ['step_embed.0.weight', 'downsampling.0.0.weight', 'downsampling.1.0.in_conv.weight',
            'downsampling.1.0.out_conv.weight', 'middle_block.1.qkv_nin.weight', 'middle_block.1.proj_out.weight',
            'class_embedding.weight', 'out.2.weight', 'upsampling.0.0.step_embedding.bias']
    save('init_seed0.npz', **{k: sd[k].numpy() for k in keys})


# ------------------------------------------------------------------------------------------------ samplers
def gen_samplers():
    for name, case in SAMPLER_CASES.items():
        cfg = dict(TINY_CFGS[case['cfg']])
        learned = case['var'] in ('learned', 'learned_interpolation')
        cfg['out_channels'] = cfg['in_channels'] * (2 if learned else 1)
        sd = UO.synth_state_dict(cfg, seed=case.get('wseed', 99), sigma_zero=case.get('sigma_zero', 0.005))
        m = ref_model(cfg, sd)
        S = case['S']
        d = Diffusion(m, 1000, S, case['var'], 'simple', beta_schedule=case['sched'],
                      guidance_method=case.get('guidance'), guidance_strength=case.get('w'),
                      use_ddim=case['ddim'], ddim_eta=case.get('eta'), device=torch.device('cpu'))
        B = 2
        R = cfg['resolution']
        C = cfg['in_channels']
        torch.manual_seed(0)
        xT = torch.randn(B, C, R, R)
        y = labels_for(cfg, B)
        kwargs = {'y': y} if y is not None else {}
        if case.get('guidance') == 'classifier_free':
            kwargs = {'y': torch.tensor([1, 2])}
            y = kwargs['y']
        # free-running loop with captured noise: patch torch.randn_like to record the draws (step order S-1..0)
        torch.manual_seed(1)
        noises = torch.randn(S, B, C, R, R)
        it = {'i': S - 1}
        orig = torch.randn_like

        def fake_randn_like(z, *a, **k):
            n = noises[it['i']]
            it['i'] -= 1
            return n.clone()
        torch.randn_like = fake_randn_like
        traj = []
        try:
            x = xT
            for tstep in reversed(range(S)):
                ts = (tstep * torch.ones(B))
                with torch.no_grad():
                    if case['ddim']:
                        x, _ = d.ddim_denoising_step(x, ts, kwargs)
                    else:
                        x, _ = d.denoising_step(x, ts, kwargs)
                traj.append(x.clone())
            # and once through the public denoise() for the same result
            it['i'] = S - 1
            out = d.denoise(x=xT, kwargs=kwargs, batch_size=B, progress=False)
        finally:
            torch.randn_like = orig
        assert torch.equal(out, traj[-1])
        # oracle check
        sch = DO.Schedule(1000, S, case['sched'])
        so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), sch, case['var'],
                              use_ddim=case['ddim'], ddim_eta=case.get('eta'),
                              guidance_method=case.get('guidance'), guidance_strength=case.get('w'))
        otraj = []
        so.denoise(xT, y, noises=noises, trace=otraj)
        e0 = (otraj[0] - traj[0]).abs().max().item()
        eN = (otraj[-1] - traj[-1]).abs().max().item()
        print('  ', name, 'oracle-vs-reference first-step', e0, 'final', eN, 'final absmax', out.abs().max().item())
        assert e0 < 1e-5 and eN < 1e-3, (name, e0, eN)
        arrs = dict(xT=xT.numpy(), noises=noises.numpy(), traj=torch.stack(traj).numpy())
        if y is not None:
            arrs['y'] = y.numpy()
        save('sampler_{}.npz'.format(name), **arrs)


# ------------------------------------------------------------------------------------------------ public per-step surface
def gen_sampler_steps():
    """The reference's public per-step methods (diffusion.py:232-369) as a caller that walks the chain itself sees them:
    for every sampler case, teacher-forced along the stored trajectory of sampler_<case>.npz (x_t of step i = x_T or the
    reference's own x after step i - 1, the stored noise draws): the (sample, pred_x0) tuple of ddim_denoising_step /
    denoising_step with clip_x=True (sample must equal the stored trajectory) and with clip_x=False, get_eps_and_log_var at
    three step indices, one step with a DIFFERENT step index per image (t = [S - 1, 0]) and diffusion_step with per-image
    indices."""
    for name, case in SAMPLER_CASES.items():
        g = np.load(os.path.join(OUT, 'sampler_{}.npz'.format(name)))
        cfg = dict(TINY_CFGS[case['cfg']])
        learned = case['var'] in ('learned', 'learned_interpolation')
        cfg['out_channels'] = cfg['in_channels'] * (2 if learned else 1)
        sd = UO.synth_state_dict(cfg, seed=case.get('wseed', 99), sigma_zero=case.get('sigma_zero', 0.005))
        m = ref_model(cfg, sd)
        S = case['S']
        d = Diffusion(m, 1000, S, case['var'], 'simple', beta_schedule=case['sched'],
                      guidance_method=case.get('guidance'), guidance_strength=case.get('w'),
                      use_ddim=case['ddim'], ddim_eta=case.get('eta'), device=torch.device('cpu'))
        xT, noises, traj = torch.from_numpy(g['xT']), torch.from_numpy(g['noises']), torch.from_numpy(g['traj'])
        B = xT.shape[0]
        kwargs = {'y': torch.from_numpy(g['y'])} if 'y' in g.files else {}
        step = d.ddim_denoising_step if case['ddim'] else d.denoising_step
        cur = {}
        orig = torch.randn_like

        def fake_randn_like(z, *a, **k):
            return cur['n'].clone()
        torch.randn_like = fake_randn_like
        try:
            pred, s_nc, p_nc = [], [], []
            for i, t in enumerate(reversed(range(S))):
                xt = xT if i == 0 else traj[i - 1]
                ts = t * torch.ones(B)
                cur['n'] = noises[t]
                with torch.no_grad():
                    smp, p0 = step(xt, ts, kwargs)
                    assert torch.equal(smp, traj[i]), (name, t)
                    pred.append(p0.float())
                    a, b = step(xt, ts, kwargs, clip_x=False)
                    s_nc.append(a.float())
                    p_nc.append(b.float())
            # one step index per image: image 0 at the head of the chain, image 1 at its masked last step
            t_mixed = torch.tensor([float(S - 1), 0.0])
            cur['n'] = noises[0]
            with torch.no_grad():
                mix_s, mix_p = step(traj[S // 2], t_mixed, kwargs)
                eidx = [S - 1, S // 2, 0]
                eps, lv = [], []
                for t in eidx:
                    e, v = d.get_eps_and_log_var(traj[S // 2], t * torch.ones(B), kwargs)
                    eps.append(e.float())
                    lv.append(v.expand(e.shape).float())
            x0 = torch.tanh(xT)
            q = d.diffusion_step(x0, t_mixed, noise=noises[1])
        finally:
            torch.randn_like = orig
        clipped = float((torch.stack(p_nc).abs() > 1).float().mean())
        print('  ', name, 'fraction of pred_x0 elements the clamp acts on', round(clipped, 3), 'noclip absmax',
              float(torch.stack(p_nc).abs().max()))
        assert clipped > 0.01, 'clip_x=False case does not differ from the clipped one'
        save('sampler_steps_{}.npz'.format(name), pred_x0=torch.stack(pred).numpy(), noclip_sample=torch.stack(s_nc).numpy(),
             noclip_pred_x0=torch.stack(p_nc).numpy(), t_mixed=t_mixed.numpy(), mixed_x=traj[S // 2].numpy(),
             mixed_sample=mix_s.float().numpy(), mixed_pred_x0=mix_p.float().numpy(), eps_indices=np.array(eidx),
             eps=torch.stack(eps).numpy(), log_var=torch.stack(lv).numpy(), q_mixed=q.float().numpy())


# ------------------------------------------------------------------------------------------------ config 1 end-to-end
def gen_config1():
    """BASELINE config[0]: EMNIST preset model, 50-step DDIM eta=0, B=4, CPU (SURVEY 8(d) Config 1)."""
    cfg = dict(ref_presets.EMNIST_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    m = ref_model(cfg, sd)
    d = Diffusion(m, 1000, 50, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=True, ddim_eta=0.0,
                  device=torch.device('cpu'))
    torch.manual_seed(0)
    xT = torch.randn(4, 1, 28, 28)
    y = (torch.arange(4) * 37) % 27
    out = d.denoise(x=xT, kwargs={'y': y}, batch_size=4, progress=False)
    sch = DO.Schedule(1000, 50, 'cosine')
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), sch, 'learned_interpolation',
                          use_ddim=True, ddim_eta=0.0)
    o = so.denoise(xT, y)
    err = (o - out).abs().max().item()
    print('   config1 oracle-vs-reference', err, 'absmax', out.abs().max().item())
    assert err < 1e-3
    u8 = ((out + 1) * 127.5).clamp(0, 255).to(torch.uint8)              # sample.py:94 + :164
    # what the reference WRITES for a 1-channel model: the same tensor expressions as sample.py:94,98-100 (float
    # inversion, stacked to 3 channels), :164 (uint8 truncation, HWC) and :170-171 (second inversion, channel 0)
    o = ((out + 1) * 127.5).clamp(0, 255).cpu()
    o3 = torch.stack((255 - o.squeeze(),) * 3, dim=1)
    saved = o3.to(torch.uint8).permute(0, 2, 3, 1).detach().numpy()
    saved = 255 - saved[..., 0]
    save('config1_emnist_ddim50.npz', xT=xT.numpy(), y=y.numpy(), out=out.numpy(), u8=u8.numpy(), u8_saved=saved)


# ------------------------------------------------------------------------------------------------ headline sizes
HEADLINE_ROWS = (0, 31, 63)


def gen_config2():
    """BASELINE configs[1] at the batch bench.py times (64x64 preset, B=64, bench.py's x_T / label recipe): the REAL
    reference on rows 0 / 31 / 63 of that batch -- one forward at t = 498 and one DDIM step at the first rescaled index
    of the 250-step cosine chain -- and the preset's OWN sampling configuration (default_args.py:15-21: 25-step DDIM,
    eta 0, cosine) free-running at B=2 with the contractive synthetic weights (sigma_zero = 0.005)."""
    cfg = dict(ref_presets.OPENAI_64_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    m = ref_model(cfg, sd)
    torch.manual_seed(0)
    x = torch.randn(64, 3, 64, 64)
    y = (torch.arange(64) * 37) % 1000
    idx = torch.tensor(HEADLINE_ROWS)
    xr, yr = x[idx], y[idx]
    t = torch.full((len(idx),), 498)
    with torch.no_grad():
        out = m(xr, t, yr)
    o = UO.unet_forward(sd, cfg, xr, t, yr)
    e = (o - out).abs().max().item()
    print('   config2 rows forward oracle-vs-reference', e, 'absmax', out.abs().max().item())
    assert e < 5e-5
    d = Diffusion(m, 1000, 250, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=True, ddim_eta=0.0,
                  device=torch.device('cpu'))
    with torch.no_grad():
        step, _ = d.ddim_denoising_step(xr, 249 * torch.ones(len(idx)), {'y': yr})
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, 250, 'cosine'),
                          'learned_interpolation', use_ddim=True, ddim_eta=0.0)
    e = (so.ddim_step(xr, 249, yr)[0] - step).abs().max().item()
    print('   config2 rows DDIM step oracle-vs-reference', e)
    assert e < 5e-5
    # the preset's own chain
    dargs = dict(ref_presets.OPENAI_64_DIFFUSION_ARGS)
    d25 = Diffusion(model=m, **dargs, device=torch.device('cpu'))
    assert d25.use_ddim and d25.rescaled_num_steps == 25
    x2, y2 = x[:2].clone(), y[:2].clone()
    keep = (0, 4, 12, 24)             # trajectory after 1, 5, 13 and all 25 steps
    traj = []
    xx = x2
    with torch.no_grad():
        for i, ts in enumerate(reversed(range(25))):
            xx, _ = d25.ddim_denoising_step(xx, ts * torch.ones(2), {'y': y2})
            if i in keep:
                traj.append(xx.clone())
        final = d25.denoise(x=x2, kwargs={'y': y2}, batch_size=2, progress=False)
    assert torch.equal(final, traj[-1])
    so25 = DO.SamplerOracle(lambda a, b, c: UO.unet_forward(sd, cfg, a, b, c), DO.Schedule(1000, 25, 'cosine'),
                            'learned_interpolation', use_ddim=True, ddim_eta=0.0)
    e = (so25.denoise(x2, y2) - final).abs().max().item()
    print('   config2 preset 25-step DDIM chain oracle-vs-reference', e, 'absmax', final.abs().max().item())
    assert e < 1e-3
    save('config2_headline_rows.npz', rows=np.array(HEADLINE_ROWS), t=t.numpy(), out=out.numpy(), ddim_first=249,
         ddim_step=step.numpy(), chain_keep=np.array(keep), chain_traj=torch.stack(traj).numpy())


def gen_config2_steps():
    """BASELINE configs[1]'s chain pinned at MORE than its first index: teacher-forced DDIM steps of the 250-step cosine
    chain at rescaled indices 125, 1 and 0 (the middle of the chain, the low-noise coefficients, and the t = 0 step whose
    noise term is masked, diffusion.py:365-366) for rows 0 / 31 / 63 of the B=64 batch.  x_t comes from the reference's own
    Diffusion.diffuse (diffusion.py:133-153) of a clean image x_0 = tanh(x) with stored noise; the rows of x_t are stored so
    that the test feeds the B=64 plan exactly what the reference saw."""
    cfg = dict(ref_presets.OPENAI_64_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    m = ref_model(cfg, sd)
    torch.manual_seed(0)
    x = torch.randn(64, 3, 64, 64)
    y = (torch.arange(64) * 37) % 1000
    idx = torch.tensor(HEADLINE_ROWS)
    x0, yr = torch.tanh(x[idx]), y[idx]
    nz = torch.randn(len(idx), 3, 64, 64, generator=torch.Generator().manual_seed(77))
    d = Diffusion(m, 1000, 250, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=True, ddim_eta=0.0,
                  device=torch.device('cpu'))
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, 250, 'cosine'),
                          'learned_interpolation', use_ddim=True, ddim_eta=0.0)
    arrs = dict(rows=np.array(HEADLINE_ROWS), indices=np.array([125, 1, 0]))
    for i in (125, 1, 0):
        xt = d.diffuse(x0, steps_to_do=i + 1, noise=nz)
        with torch.no_grad():
            step, pred = d.ddim_denoising_step(xt, i * torch.ones(len(idx)), {'y': yr})
        e = (so.ddim_step(xt, i, yr)[0] - step).abs().max().item()
        print('   config2 rows DDIM step at index', i, 'oracle-vs-reference', e, 'absmax', step.abs().max().item())
        assert e < 5e-5
        arrs['xt_%d' % i] = xt.numpy()
        arrs['step_%d' % i] = step.numpy()
        arrs['pred_x0_%d' % i] = pred.float().numpy()          # round 6: the second element of the reference's tuple
    save('config2_headline_steps.npz', **arrs)


def gen_large_rows():
    """Two rows of the full-batch forwards of BASELINE configs[3] (128x128 preset + null class, 2B = 32 forwards per step,
    second half = the null class) and configs[4] (256x256 preset, B = 16) through the REAL reference in fp32.  Stored as a
    strided sub-sample of the output plus its mean / mean |.| (the full tensors would be 0.8 / 3.1 MB)."""
    for name, pre, ncls, NB, stride, rows in (('config4', 'OPENAI_128_MODEL_ARGS', 1001, 32, 2, (0, 31)),
                                              ('config5', 'OPENAI_256_MODEL_ARGS', 1000, 16, 4, (0, 15))):
        cfg = dict(getattr(ref_presets, pre))
        cfg['num_classes'] = ncls
        sd = UO.synth_state_dict(cfg, seed=1234)
        m = ref_model(cfg, sd)
        R = cfg['resolution']
        torch.manual_seed(0)
        B = NB // 2 if name == 'config4' else NB
        x = torch.randn(B, 3, R, R)
        y = (torch.arange(B) * 37) % 1000 + (1 if name == 'config4' else 0)
        if name == 'config4':             # the classifier-free batch: [x | x], [y | null class 0]
            x, y = torch.cat([x, x]), torch.cat([y, torch.zeros_like(y)])
        idx = torch.tensor(rows)
        t = torch.full((len(rows),), 321)
        with torch.no_grad():
            out = m(x[idx], t, y[idx])
        o = UO.unet_forward(sd, cfg, x[idx], t, y[idx])
        e = (o - out).abs().max().item()
        print('  ', name, 'rows forward oracle-vs-reference', e, 'absmax', out.abs().max().item())
        assert e < 1e-4
        save('{}_fullbatch_rows.npz'.format(name), rows=np.array(rows), t=t.numpy(), stride=stride,
             out_sub=out[:, :, ::stride, ::stride].numpy(), mean=out.mean(dim=(1, 2, 3)).numpy(),
             absmean=out.abs().mean(dim=(1, 2, 3)).numpy())
        del m, sd


# ------------------------------------------------------------------------------------------------ N3: diffuse / img2img
def gen_diffuse():
    """Reference Diffusion.diffuse (diffusion.py:133-153,232-240) and the --start_img chain of sample.py:54-64,76-78:
    diffuse(x_0, k) -> denoise(steps_to_do=k), DDIM eta=0 and DDPM, with the reverse-step noise captured."""
    cfg = dict(TINY_CFGS['adagn_updown'])
    sd = UO.synth_state_dict(cfg, seed=1234)
    m = ref_model(cfg, sd)
    torch.manual_seed(3)
    x0, nz = torch.randn(2, 3, 16, 16).clamp(-1, 1), torch.randn(2, 3, 16, 16)
    y = torch.tensor([1, 7])
    arrs = dict(x0=x0.numpy(), nz=nz.numpy(), y=y.numpy())
    d = Diffusion(m, 1000, 10, 'learned_interpolation', 'hybrid', beta_schedule='cosine', device=torch.device('cpu'))
    for steps in (1, 4, 10, None, 99):
        arrs['diffuse/{}'.format(steps)] = d.diffuse(x0, steps_to_do=steps, noise=nz).numpy()
    torch.manual_seed(5)
    noises = torch.randn(10, 2, 3, 16, 16)
    arrs['noises'] = noises.numpy()
    orig = torch.randn_like
    for use_ddim in (True, False):
        kw = dict(use_ddim=True, ddim_eta=0.0) if use_ddim else dict(use_ddim=False)
        d = Diffusion(m, 1000, 10, 'learned_interpolation', 'hybrid', beta_schedule='cosine',
                      device=torch.device('cpu'), **kw)
        so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, 10, 'cosine'),
                              'learned_interpolation', **kw)
        for k in (1, 4, 10):
            it = {'i': k - 1}

            def fake_randn_like(z, *a, **kk):
                n = noises[it['i']]
                it['i'] -= 1
                return n.clone()
            torch.randn_like = fake_randn_like
            try:
                xk = d.diffuse(x_0=x0, steps_to_do=k, noise=nz)
                out = d.denoise(x=xk, kwargs={'y': y}, batch_size=2, progress=False, steps_to_do=k)
            finally:
                torch.randn_like = orig
            ref = so.denoise(so.diffuse(x0, k, nz), y, steps_to_do=k, noises=list(noises))
            err = (ref - out).abs().max().item()
            print('   img2img', 'ddim' if use_ddim else 'ddpm', k, 'oracle-vs-reference', err)
            assert err < 1e-4
            arrs['chain/{}/{}'.format('ddim' if use_ddim else 'ddpm', k)] = out.numpy()
    save('diffuse_img2img.npz', **arrs)


# ------------------------------------------------------------------------------------------------ CLI dicts (A12)
def gen_cli():
    res = {}
    for name, argv in CLI_CASES.items():
        parser = make_argparser('diff_sample')
        args = parser.parse_args(argv)
        other, margs, dargs = get_dicts_from_args(args)
        res[name] = dict(argv=argv, other=other, model=margs, diff=dargs)
    with open(os.path.join(OUT, 'cli_dicts.json'), 'w') as f:
        json.dump(res, f, indent=1, default=lambda o: list(o))
    print('wrote cli_dicts.json', list(res))


# ------------------------------------------------------------------------------------------------ embedding known answers
def gen_embed():
    from nicediffusion.model import timestep_embedding
    t = torch.tensor([0, 2, 10, 498, 998])
    save('timestep_embedding.npz', t=t.numpy(), e192=timestep_embedding(t, 192).numpy(),
         e64=timestep_embedding(t, 64).numpy(), e33=timestep_embedding(t, 33).numpy())


if __name__ == '__main__':
    which = sys.argv[1:] or ['schedules', 'embed', 'tiny', 'init', 'samplers', 'sampler_steps', 'cli', 'config1', 'presets', 'diffuse', 'config2', 'config2_steps', 'large_rows']
    fns = dict(sampler_steps=gen_sampler_steps, config2_steps=gen_config2_steps, diffuse=gen_diffuse, schedules=gen_schedules, embed=gen_embed, tiny=gen_tiny_forwards, init=gen_init,
               samplers=gen_samplers, cli=gen_cli, config1=gen_config1, presets=gen_presets, config2=gen_config2,
               large_rows=gen_large_rows)
    for w in which:
        print('==', w)
        fns[w]()
