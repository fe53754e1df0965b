#!/usr/bin/env python3
"""What would Winograd F(4x4,3x3) in fp32 do to parity?  CPU-only study (no GPU, no product code): the oracle's 3x3
convolutions are swapped for fp32 emulations of F(2x2,3x3) (what the HIP kernels compute) and F(4x4,3x3), and a forward
plus a short free-running DDIM chain are compared with the plain oracle.  Test-infrastructure side only.

    python tools/f4_numerics.py            # tiny 32x32 model, forward + 50-step chain
    python tools/f4_numerics.py preset64   # the 64x64 preset against the REAL reference's goldens
                                           # (tests/golden/config2_headline_rows.npz): forward rows + the preset's own
                                           # 25-step DDIM chain, F(2x2) / F(4x4) everywhere / F(4x4) on the 64x64 and 32x32
                                           # levels only (the table in DESIGN.md section 6)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import unet_oracle as UO, diffusion_oracle as DO

_T = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                      [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)),
}
_real_conv2d = F.conv2d
MODE = [0]          # 0 = plain, 2 = F(2x2,3x3), 4 = F(4x4,3x3)
F4_MIN_HW = [0]     # F(4x4) only on maps at least this large (smaller ones fall back to F(2x2)): the "mixed" assignment


def wino_conv(x, w, b, m):
    Bt, G, At = (t.float() for t in _T[m])
    a = m + 2
    Bn, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    U = torch.einsum('ai,ncij,bj->ncab', G, w, G)
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                    # [B, C, th, tw, a, a]
    V = torch.einsum('ai,bcyxij,kj->bcyxak', Bt, tiles, Bt)
    M = torch.einsum('ncak,bcyxak->bnyxak', U, V)
    Y = torch.einsum('ia,bnyxak,jk->bnyxij', At, M, At)
    out = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], H, W)
    return out if b is None else out + b[None, :, None, None]


def patched_conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    m = MODE[0]
    if m == 4 and x.shape[-1] < F4_MIN_HW[0]:
        m = 2
    if (m and w.shape[-1] == 3 and stride == 1 and padding == 1 and x.shape[-1] % m == 0 and x.shape[-2] % m == 0
            and x.dtype == torch.float32):
        return wino_conv(x, w, b, m)
    return _real_conv2d(x, w, b, stride, padding, *a, **k)


UO.F.conv2d = patched_conv2d            # the oracle module's F is torch.nn.functional itself: patched for this process


def main():
    cfg = dict(resolution=32, in_channels=3, model_channels=64, out_channels=6, num_res_blocks=2,
               attention_resolutions=(8,), channel_mult=(1, 2, 2), num_head_channels=32, num_classes=10,
               use_adaptive_gn=True, resblock_updown=True, split_qkv_first=True, dropout=0.0, num_heads=4)
    sd = UO.synth_state_dict(cfg, seed=11)
    torch.manual_seed(0)
    x = torch.randn(2, 3, 32, 32)
    t = torch.tensor([500, 20])
    y = torch.tensor([1, 7])
    MODE[0] = 0
    ref = UO.unet_forward(sd, cfg, x, t, y)
    print('forward, 32x32 tiny model (all three levels 32/16/8 are multiples of 4); deviation from the plain fp32 forward:')
    for m, name in ((2, 'F(2x2,3x3) fp32'), (4, 'F(4x4,3x3) fp32')):
        MODE[0] = m
        out = UO.unet_forward(sd, cfg, x, t, y)
        e = (out - ref).abs()
        print('  %-18s max %.3e   rms %.3e   (output absmax %.3f)' % (name, e.max().item(), e.pow(2).mean().sqrt().item(),
                                                                     ref.abs().max().item()))
    S = 50
    sch = DO.Schedule(1000, S, 'cosine')
    outs = {}
    for m in (0, 2, 4):
        MODE[0] = m
        so = DO.SamplerOracle(lambda a, b_, c: UO.unet_forward(sd, cfg, a, b_, c), sch, 'learned_interpolation',
                              use_ddim=True, ddim_eta=0.0)
        outs[m] = so.denoise(x.clone(), y)
    print('%d-step free-running DDIM chain; max |x_0 - x_0(plain fp32)| (tolerance of the path: 1e-3):' % S)
    for m, name in ((2, 'F(2x2,3x3)'), (4, 'F(4x4,3x3)')):
        print('  %-12s %.3e' % (name, (outs[m] - outs[0]).abs().max().item()))
    MODE[0] = 0


def preset64():
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
    from nicediffusion import default_args as DA
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'config2_headline_rows.npz'))
    cfg = dict(DA.OPENAI_64_MODEL_ARGS)
    sd = UO.synth_state_dict(cfg, seed=1234)
    torch.manual_seed(0)
    x = torch.randn(64, 3, 64, 64)
    y = (torch.arange(64) * 37) % 1000
    rows = torch.from_numpy(g['rows'])
    t = torch.from_numpy(g['t'])
    modes = ((0, 0, 'plain fp32'), (2, 0, 'F(2x2) everywhere'), (4, 0, 'F(4x4) everywhere'), (4, 32, 'F(4x4) on 64/32, F(2x2) on 16/8'),
             (4, 16, 'F(4x4) on 64/32/16, F(2x2) on 8'))
    print('64x64 preset (BASELINE configs[1] weights), deviation from the REAL reference (tests/golden/config2_headline_rows.npz)')
    print('forward, rows 0/31/63 at t=498 (output absmax %.3f):' % float(np.abs(g['out']).max()))
    for m, mn, name in modes:
        MODE[0], F4_MIN_HW[0] = m, mn
        out = UO.unet_forward(sd, cfg, x[rows], t, y[rows])
        print('  %-36s max %.3e' % (name, float((out - torch.from_numpy(g['out'])).abs().max())), flush=True)
    keep = [int(k) for k in g['chain_keep']]
    print("the preset's own 25-step DDIM chain (free-running, B=2), max |x - reference| after 1 / 5 / 13 / 25 steps:")
    sch = DO.Schedule(1000, 25, 'cosine')
    for m, mn, name in modes:
        MODE[0], F4_MIN_HW[0] = m, mn
        so = DO.SamplerOracle(lambda a, b_, c: UO.unet_forward(sd, cfg, a, b_, c), sch, 'learned_interpolation',
                              use_ddim=True, ddim_eta=0.0)
        xx = x[:2].clone()
        errs = []
        for i, ts in enumerate(reversed(range(25))):
            xx = so.ddim_step(xx, ts, y[:2])[0]
            if i in keep:
                errs.append(float((xx - torch.from_numpy(g['chain_traj'][keep.index(i)])).abs().max()))
        print('  %-36s %s' % (name, ' / '.join('%.2e' % e for e in errs)), flush=True)
    MODE[0] = 0


if __name__ == '__main__':
    preset64() if len(sys.argv) > 1 and sys.argv[1] == 'preset64' else main()
