#!/usr/bin/env python3
"""CPU-only go / no-go for a bf16 Winograd F(2x2,3x3) convolution in the bf16 plans (BASELINE configs[3] / [4]).

What a bf16 Winograd kernel would compute: the input transform B^T d B in fp32 on the bf16 activations, ROUNDED TO BF16
(the MFMA operand), weights U = G g G^T formed in float64 and rounded to bf16 once, products accumulated in fp32
(v_mfma_f32_*_bf16), the output transform A^T M A in fp32, bias, the result stored as bf16.  Against it, the arithmetic of
today's direct kernels emulated the same way: bf16 operands, fp32 accumulation, bf16 stores.  Both are put against the
REAL reference's fp32 rows of the full-batch forwards (tests/golden/config4_fullbatch_rows.npz, config5_fullbatch_rows.npz
-- the vectors tests/test_gpu_bf16.py holds the bf16 plans to; measured there on MI355X: rel. rms 9.6e-3 / 1.14e-2 and
8.9e-3 / 9.5e-3).  Decision rule: price a kernel only if the Winograd plan stays within 2 x the
direct plan's relative rms.  Test-infrastructure side only: the oracle with its convolutions patched, no product code.

    python tools/bf16_wino_numerics.py [config4] [config5]      -> profiles/r06_bf16_winograd_numerics.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import unet_oracle as UO

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
_conv2d, _conv1d = F.conv2d, F.conv1d
MODE = ['fp32']        # 'fp32' | 'direct' (bf16 operands, fp32 accumulate, bf16 store) | 'wino' (F(2x2,3x3) on the 3x3 layers)
WINO_MIN_C = [0]       # Winograd only on layers with at least this many input channels (the first conv has 3)
STATS = {}


def rb(t):
    return t.to(torch.bfloat16).float()


def wino_bf16(x, w, b):
    """x: bf16-valued fp32 [B, C, H, W]; w fp32 [N, C, 3, 3].  Returns the fp32 accumulators + bias."""
    Bn, C, H, W = x.shape
    N = w.shape[0]
    U = rb(torch.einsum('ai,ncij,bj->abnc', G, w.double(), G).float()).reshape(16, N, C)       # float64 transform, rounded once
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                                 # [B, C, th, tw, 4, 4]
    th, tw = tiles.shape[2], tiles.shape[3]
    V = torch.einsum('ai,bcyxij,kj->akcbyx', BT, tiles, BT)                                    # fp32 transform ...
    amax = float(V.abs().max())
    V = rb(V).reshape(16, C, Bn * th * tw)                                                     # ... rounded to the MFMA operand
    M = torch.bmm(U, V).reshape(4, 4, N, Bn, th, tw)                                           # fp32 accumulation
    Y = torch.einsum('ia,aknbyx,jk->bnyixj', AT, M, AT).reshape(Bn, N, H, W)
    STATS.setdefault('v_over_x', []).append(amax / max(float(x.abs().max()), 1e-30))
    return Y if b is None else Y + b[None, :, None, None]


def conv2d_patched(x, w, b=None, stride=1, padding=0, *a, **k):
    if MODE[0] == 'fp32':
        return _conv2d(x, w, b, stride, padding, *a, **k)
    last = w.shape[0] <= 8                                   # the UNet's last convolution writes its fp32 accumulators
    xq = rb(x)
    if (MODE[0] == 'wino' and w.shape[-1] == 3 and stride == 1 and padding == 1 and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0
            and w.shape[1] >= WINO_MIN_C[0] and not last):
        out = wino_bf16(xq, w, b)
    else:
        out = _conv2d(xq, rb(w), b, stride, padding, *a, **k)
    return out if last else rb(out)


def conv1d_patched(x, w, b=None, *a, **k):
    if MODE[0] == 'fp32':
        return _conv1d(x, w, b, *a, **k)
    return rb(_conv1d(rb(x), rb(w), b, *a, **k))


UO.F.conv2d = conv2d_patched
UO.F.conv1d = conv1d_patched


def errs(got, ref):
    d = got - ref
    return float(np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean())), float(np.abs(d).max() / np.abs(ref).max())


def run(name):
    from nicediffusion import default_args as DA
    g = np.load(os.path.join(ROOT, 'tests', 'golden', '{}_fullbatch_rows.npz'.format(name)))
    cfg = dict(DA.OPENAI_128_MODEL_ARGS if name == 'config4' else DA.OPENAI_256_MODEL_ARGS)
    B = 16
    if name == 'config4':
        cfg['num_classes'] += 1
    sd = UO.synth_state_dict(cfg, seed=1234)
    R = cfg['resolution']
    torch.manual_seed(0)
    x = torch.randn(B, 3, R, R)
    y = (torch.arange(B) * 37) % 1000 + (1 if name == 'config4' else 0)
    if name == 'config4':
        x, y = torch.cat([x, x]), torch.cat([y, torch.zeros_like(y)])
    rows = torch.from_numpy(g['rows'])
    t = torch.from_numpy(g['t'])
    st = int(g['stride'])
    print('{}: {}x{} preset, rows {} of the full forward batch, t = {}; deviation from the REAL reference (fp32), relative rms / max over '
          'absmax:'.format(name, R, R, list(g['rows']), int(t[0])), flush=True)
    res = {}
    for mode, minc, label in (('fp32', 0, 'oracle fp32 (sanity: the restatement)'), ('direct', 0, "bf16 direct (today's arithmetic)"),
                              ('wino', 16, 'bf16 Winograd F(2x2,3x3) on every 3x3 layer but the first / last'),
                              ('wino', 1024, 'bf16 Winograd F(2x2,3x3) on the 3x3 layers with >= 1024 input channels only')):
        MODE[0], WINO_MIN_C[0] = mode, minc
        STATS.clear()
        outs = []
        for r in rows:                                      # rows are independent: one at a time keeps the emulation's memory small
            with torch.no_grad():
                outs.append(UO.unet_forward(sd, cfg, x[r:r + 1], t[:1], y[r:r + 1]))
        out = torch.cat(outs).numpy()[:, :, ::st, ::st]
        res[label] = errs(out, g['out_sub'])
        extra = ''
        if STATS.get('v_over_x'):
            extra = '   (max |B^T d B| / max |d| over the layers: {:.2f})'.format(max(STATS['v_over_x']))
        print('  {:78s} {:.3e} / {:.3e}{}'.format(label, res[label][0], res[label][1], extra), flush=True)
    MODE[0] = 'fp32'
    d = res["bf16 direct (today's arithmetic)"][0]
    w = res['bf16 Winograd F(2x2,3x3) on every 3x3 layer but the first / last'][0]
    print('  -> Winograd / direct = {:.2f} x the relative rms (go if <= 2)'.format(w / d), flush=True)


if __name__ == '__main__':
    torch.set_num_threads(8)
    for nm in (sys.argv[1:] or ['config4', 'config5']):
        run(nm)
