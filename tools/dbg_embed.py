import sys, os, math
sys.path.insert(0, 'nice-diffusion_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from nicediffusion.model import timestep_embedding
g = np.load('tests/golden/timestep_embedding.npz')
t = torch.from_numpy(g['t']).cuda()
got = timestep_embedding(t, 192).cpu().numpy()
err = np.abs(got - g['e192'])
idx = np.argwhere(err > 1e-6)
half = 96
f = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / half)).numpy()
for b, i in idx[:20]:
    arg = np.float32(g['t'][b]) * f[i % half]
    print(b, i, 'arg', arg, 'got', got[b, i], 'ref', g['e192'][b, i], 'torch.cuda', (torch.cos if i < half else torch.sin)(torch.tensor([arg]).cuda()).item())
