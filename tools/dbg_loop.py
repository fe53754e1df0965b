import sys, os
sys.path.insert(0, 'nice-diffusion_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from nicediffusion.diffusion import Diffusion
from nicediffusion.model import DiffusionModel
from oracle import unet_oracle as UO
from tests.cases import TINY_CFGS, SAMPLER_CASES
DEV = torch.device('cuda')
for name in sorted(SAMPLER_CASES):
    g = np.load('tests/golden/sampler_{}.npz'.format(name))
    case = SAMPLER_CASES[name]
    cfg = dict(TINY_CFGS[case['cfg']])
    learned = case['var'] in ('learned', 'learned_interpolation')
    cfg['out_channels'] = cfg['in_channels'] * (2 if learned else 1)
    m = DiffusionModel(**cfg); m.load_state_dict(UO.synth_state_dict(cfg, seed=99)); m.to(DEV).eval()
    S = case['S']
    d = Diffusion(m, 1000, S, case['var'], 'simple', beta_schedule=case['sched'], guidance_method=case.get('guidance'),
                  guidance_strength=case.get('w'), use_ddim=case['ddim'], ddim_eta=case.get('eta'), device=DEV)
    y = torch.from_numpy(g['y']).to(DEV) if 'y' in g.files else None
    kwargs = {'y': y} if y is not None else None
    tr = []
    d.denoise(x=torch.from_numpy(g['xT']), kwargs=kwargs, batch_size=2, progress=False, noise=torch.from_numpy(g['noises']), trace=tr)
    errs = [float(np.abs(tr[i].cpu().numpy() - g['traj'][i]).max()) for i in range(S)]
    print(name, ' '.join('%.1e' % e for e in errs))
