"""The CPU oracle against the golden vectors produced by the real reference (tools/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as UO
from oracle import diffusion_oracle as DO
from tests.cases import TINY_CFGS, SCHEDULE_CASES, SAMPLER_CASES

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('name', sorted(TINY_CFGS))
def test_tiny_forward_and_taps(golden_dir, name):
    g = _load(golden_dir, 'fwd_{}.npz'.format(name))
    cfg = TINY_CFGS[name]
    sd = UO.synth_state_dict(cfg, seed=1234)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    taps = {}
    out = UO.unet_forward(sd, cfg, torch.from_numpy(g['x']), torch.from_numpy(g['t']), y, taps=taps)
    assert np.abs(out.numpy() - g['out']).max() < 1e-5
    assert np.abs(g['out']).max() > 0.05, 'vacuous fixture (zero-init trap)'
    n = 0
    for k in g.files:
        if k.startswith('tap/'):
            assert np.abs(taps[k[4:]].numpy() - g[k]).max() < 1e-5, k
            n += 1
    assert n >= 10


def test_preset_emnist_forward(golden_dir):
    g = _load(golden_dir, 'fwd_preset_emnist.npz')
    from nicediffusion.default_args import EMNIST_MODEL_ARGS
    cfg = dict(EMNIST_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    out = UO.unet_forward(sd, cfg, torch.from_numpy(g['x']), torch.from_numpy(g['t']), torch.from_numpy(g['y']))
    assert np.abs(out.numpy() - g['out']).max() < 2e-5


def test_preset_64_forward(golden_dir):
    g = _load(golden_dir, 'fwd_preset_64.npz')
    from nicediffusion.default_args import OPENAI_64_MODEL_ARGS
    cfg = dict(OPENAI_64_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    out = UO.unet_forward(sd, cfg, torch.from_numpy(g['x']), torch.from_numpy(g['t']), torch.from_numpy(g['y']))
    assert np.abs(out.numpy() - g['out']).max() < 1e-4


def test_param_shapes_match_reference_state_dicts(golden_dir):
    import json
    from nicediffusion import default_args as DA
    meta = json.load(open(os.path.join(golden_dir, 'preset_state_dicts.json')))
    for pname, margs in (('EMNIST', DA.EMNIST_MODEL_ARGS), ('OPENAI_64', DA.OPENAI_64_MODEL_ARGS),
                         ('OPENAI_128', DA.OPENAI_128_MODEL_ARGS), ('OPENAI_256', DA.OPENAI_256_MODEL_ARGS)):
        shapes = UO.param_shapes(dict(margs))
        assert list(shapes) == meta[pname]['keys']
        assert [list(v) for v in shapes.values()] == meta[pname]['shapes']
    assert meta['OPENAI_64']['n_tensors'] == 541 and meta['EMNIST']['n_tensors'] == 309
    assert meta['OPENAI_64']['n_params'] == 295904454


@pytest.mark.parametrize('name', sorted(SCHEDULE_CASES))
def test_schedule_tables(golden_dir, name):
    g = _load(golden_dir, 'schedules.npz')
    T, S, sched = SCHEDULE_CASES[name]
    s = DO.Schedule(T, S, sched)
    for attr in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
                 'sqrt_one_minus_alphas_cumprod', 'sqrt_reciprocal_alphas_cumprod',
                 'sqrt_reciprocal_alphas_minus_one_cumprod', 'posterior_mean_coef_x0', 'posterior_mean_coef_xt',
                 'posterior_variance', 'log_posterior_var_clipped', 'timestep_map'):
        ref = g['{}/{}'.format(name, attr)]
        got = getattr(s, attr)
        assert got.shape == ref.shape, (name, attr)
        assert np.array_equal(got, ref), (name, attr)      # float64 bit-exact


def test_schedule_known_answers():
    """SURVEY.md 8(a) A1 probe values."""
    s = DO.Schedule(1000, 250, 'cosine')
    assert list(s.timestep_map[:3]) == [2, 6, 10] and s.timestep_map[-1] == 998
    assert s.betas[0] == 1.3841909524869855e-4
    assert s.betas[249] == 0.9599992229055568
    assert s.alphas_cumprod[249] == 2.4287669070348544e-6
    assert s.log_posterior_var_clipped[0] == -9.322145735186036
    s = DO.Schedule(1000, 50, 'linear')
    assert list(s.timestep_map[:3]) == [10, 30, 50]
    assert s.betas[0] == 2.19342749215401e-3 and s.betas[49] == 0.32735320663781853


def test_timestep_embedding_known_answers(golden_dir):
    g = _load(golden_dir, 'timestep_embedding.npz')
    t = torch.from_numpy(g['t'])
    for dim, key in ((192, 'e192'), (64, 'e64'), (33, 'e33')):
        assert np.array_equal(UO.timestep_embedding(t, dim).numpy(), g[key])
    e = UO.timestep_embedding(torch.tensor([2]), 192)[0]
    assert abs(e[0].item() - (-0.416146844625473)) < 1e-7 and abs(e[96].item() - 0.9092974066734314) < 1e-7


@pytest.mark.parametrize('name', sorted(SAMPLER_CASES))
def test_sampler_trajectories(golden_dir, name):
    g = _load(golden_dir, 'sampler_{}.npz'.format(name))
    case = SAMPLER_CASES[name]
    cfg = dict(TINY_CFGS[case['cfg']])
    learned = case['var'] in ('learned', 'learned_interpolation')
    cfg['out_channels'] = cfg['in_channels'] * (2 if learned else 1)
    sd = UO.synth_state_dict(cfg, seed=case.get('wseed', 99), sigma_zero=case.get('sigma_zero', 0.005))
    sch = DO.Schedule(1000, case['S'], case['sched'])
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), sch, case['var'],
                          use_ddim=case['ddim'], ddim_eta=case.get('eta'), guidance_method=case.get('guidance'),
                          guidance_strength=case.get('w'))
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    noises = torch.from_numpy(g['noises'])
    traj = g['traj']
    S = case['S']
    # teacher-forced: feed the reference's x_t, compare x_{t-1}
    x = torch.from_numpy(g['xT'])
    for i, t in enumerate(reversed(range(S))):
        step = so.ddim_step if case['ddim'] else so.ddpm_step
        nxt, _ = step(x, t, y, noises[t])
        assert np.abs(nxt.numpy() - traj[i]).max() < 1e-5, (name, t)
        x = torch.from_numpy(traj[i])
    # free-running
    out = so.denoise(torch.from_numpy(g['xT']), y, noises=noises)
    assert np.abs(out.numpy() - traj[-1]).max() < 1e-3


@pytest.mark.parametrize('name', sorted(SAMPLER_CASES))
def test_per_step_surface_pred_x0_clip_and_log_var(golden_dir, name):
    """The (sample, pred_x0) tuples of the reference's public per-step methods (diffusion.py:266-369), clip_x=False,
    get_eps_and_log_var and per-image step indices (tests/golden/sampler_steps_<case>.npz, tools/gen_golden.py
    gen_sampler_steps): the oracle restates them (rows are independent, so a per-image index is one call per image)."""
    g = _load(golden_dir, 'sampler_{}.npz'.format(name))
    gs = _load(golden_dir, 'sampler_steps_{}.npz'.format(name))
    case = SAMPLER_CASES[name]
    cfg = dict(TINY_CFGS[case['cfg']])
    learned = case['var'] in ('learned', 'learned_interpolation')
    cfg['out_channels'] = cfg['in_channels'] * (2 if learned else 1)
    sd = UO.synth_state_dict(cfg, seed=case.get('wseed', 99), sigma_zero=case.get('sigma_zero', 0.005))
    S = case['S']
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, S, case['sched']), case['var'],
                          use_ddim=case['ddim'], ddim_eta=case.get('eta'), guidance_method=case.get('guidance'),
                          guidance_strength=case.get('w'))
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    noises, traj, xT = torch.from_numpy(g['noises']), torch.from_numpy(g['traj']), torch.from_numpy(g['xT'])
    step = so.ddim_step if case['ddim'] else so.ddpm_step
    rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    # pred_x0 = c_t x - c'_t eps (diffusion.py:287-288) with c'_t up to 404 at the head of a 10-step chain: a 1e-7 difference in
    # eps (the CPU convolutions of a batch of 1 and of 2 differ by that) is 4e-5 in pred_x0 -- the bound scales with c'_t
    ptol = lambda t: 1e-5 + 5e-7 * float(so.s.sqrt_reciprocal_alphas_minus_one_cumprod[t])
    for i, t in enumerate(reversed(range(S))):
        xt = xT if i == 0 else traj[i - 1]
        _, p0 = step(xt, t, y, noises[t])
        assert np.abs(p0.numpy() - gs['pred_x0'][i]).max() < ptol(t), (name, t)
        a, b = step(xt, t, y, noises[t], clip_x=False)
        assert rel(a.numpy(), gs['noclip_sample'][i]) < 1e-5 and rel(b.numpy(), gs['noclip_pred_x0'][i]) < 1e-5, (name, t)
    xm = torch.from_numpy(gs['mixed_x'])
    for b, t in enumerate(int(v) for v in gs['t_mixed']):
        yy = None if y is None else y[b:b + 1]
        smp, p0 = step(xm[b:b + 1], t, yy, noises[0][b:b + 1])
        assert rel(smp.numpy(), gs['mixed_sample'][b:b + 1]) < 1e-5
        assert np.abs(p0.numpy() - gs['mixed_pred_x0'][b:b + 1]).max() < ptol(t), (name, t)
        q = so.diffuse(torch.tanh(xT[b:b + 1]), t + 1, noises[1][b:b + 1])
        assert rel(q.numpy(), gs['q_mixed'][b:b + 1]) < 1e-6
    for k, t in enumerate(int(v) for v in gs['eps_indices']):
        e, lv = so.eps_and_log_var(xm, t, y)
        assert rel(e.numpy(), gs['eps'][k]) < 1e-5 and rel(lv.numpy(), gs['log_var'][k]) < 1e-5, (name, t)


def test_config1_end_to_end(golden_dir):
    """BASELINE configs[0]: EMNIST preset, 50-step DDIM, B=4 on the CPU."""
    g = _load(golden_dir, 'config1_emnist_ddim50.npz')
    from nicediffusion.default_args import EMNIST_MODEL_ARGS
    cfg = dict(EMNIST_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, 50, 'cosine'),
                          'learned_interpolation', use_ddim=True, ddim_eta=0.0)
    out = so.denoise(torch.from_numpy(g['xT']), torch.from_numpy(g['y']))
    assert np.abs(out.numpy() - g['out']).max() < 1e-3
    u8 = ((out + 1) * 127.5).clamp(0, 255).to(torch.uint8).numpy()
    assert (np.abs(u8.astype(int) - g['u8'].astype(int)) <= 1).all()


def test_diffuse_and_img2img_chain(golden_dir):
    """N3 entry pinned by the reference: Diffusion.diffuse for steps 1/4/10/None/99 and diffuse -> denoise(steps_to_do=k)
    chains (diffusion.py:133-153,192-197,232-240; sample.py:54-64,76-78)."""
    g = _load(golden_dir, 'diffuse_img2img.npz')
    cfg = TINY_CFGS['adagn_updown']
    sd = UO.synth_state_dict(cfg, seed=1234)
    x0, nz, y = torch.from_numpy(g['x0']), torch.from_numpy(g['nz']), torch.from_numpy(g['y'])
    noises = torch.from_numpy(g['noises'])
    for use_ddim in (True, False):
        kw = dict(use_ddim=True, ddim_eta=0.0) if use_ddim else dict(use_ddim=False)
        so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, cfg, xx, tt, yy), DO.Schedule(1000, 10, 'cosine'),
                              'learned_interpolation', **kw)
        for steps in (1, 4, 10, None, 99):
            assert np.abs(so.diffuse(x0, steps, nz).numpy() - g['diffuse/{}'.format(steps)]).max() == 0
        for k in (1, 4, 10):
            out = so.denoise(so.diffuse(x0, k, nz), y, steps_to_do=k, noises=list(noises))
            assert np.abs(out.numpy() - g['chain/{}/{}'.format('ddim' if use_ddim else 'ddpm', k)]).max() < 1e-5


def test_saved_bytes_of_one_channel_model(golden_dir):
    """The bytes the reference writes for a 1-channel model are 255 - uint8(255 - v) = ceil(v), one level above the plain
    truncation for every non-integer v (sample.py:98-100,164,170-171)."""
    g = _load(golden_dir, 'config1_emnist_ddim50.npz')
    v = ((torch.from_numpy(g['out']) + 1) * 127.5).clamp(0, 255)
    saved = 255 - (255 - v).to(torch.uint8).numpy()[:, 0]
    assert np.array_equal(saved, g['u8_saved'])
    nonint = (v != v.round()).numpy()[:, 0]
    assert np.array_equal(saved[nonint].astype(int), g['u8'][:, 0][nonint].astype(int) + 1)
