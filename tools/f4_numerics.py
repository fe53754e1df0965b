#!/usr/bin/env python3
"""What would Winograd F(4x4,3x3) in fp32 do to parity?  CPU-only study (no GPU, no product code): the oracle's 3x3
convolutions are swapped for fp32 emulations of F(2x2,3x3) (what the HIP kernels compute) and F(4x4,3x3), and a forward
plus a short free-running DDIM chain are compared with the plain oracle.  Test-infrastructure side only.

    python tools/f4_numerics.py            # tiny 32x32 model, forward + 50-step chain"""
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


if __name__ == '__main__':
    main()
