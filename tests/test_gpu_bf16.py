"""bf16 path (BASELINE configs[3], [4]) on a real MI355X, through the C ABI.

The reference has no bf16 arithmetic (utils.py:83 parses --use_fp16 and never reads it; model.py:517 is fp32), so the
bar is stated here and in DESIGN.md:
  * kernel level: each bf16 kernel against a float64 CPU statement of the same op on the SAME bf16-rounded operands --
    what remains is fp32 accumulation order and the final rounding of the output to bf16 (half an ulp = 2^-9 relative);
  * model level: the bf16 forward against the reference's fp32 goldens / the fp32 HIP forward of the same weights:
    relative rms error <= 3.5e-2 and max error <= 8e-2 of the output's max magnitude per forward (every activation is
    rounded to 8 significant bits ~50 times along the deepest path: measured 1.4e-2 .. 2.4e-2 rms); a teacher-forced
    sampler step moves x_{t-1} by <= 3e-2 (x is O(1)).  Measured values are printed by the tests and recorded in DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nicediffusion import _hip
from nicediffusion import default_args as DA
from nicediffusion.diffusion import Diffusion
from nicediffusion.model import DiffusionModel
from oracle import unet_oracle as UO
from oracle import diffusion_oracle as DO
from tests.cases import TINY_CFGS

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BF = torch.bfloat16


def st():
    return torch.cuda.current_stream().cuda_stream


def lib():
    return _hip.load()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(x):
    """bf16-rounded copy (round to nearest even, as the kernels' conversions do), still fp32."""
    return x.to(BF).float()


def nhwc_bf(x, ld=None):       # [B,C,H,W] cpu fp32 -> flat NHWC bf16 on the device (channels padded to ld with garbage-free zeros)
    B, C, H, W = x.shape
    ld = C if ld is None else ld
    t = torch.zeros(B, H, W, ld, dtype=BF)
    t[..., :C] = x.permute(0, 2, 3, 1).to(BF)
    return t.contiguous().to(DEV)


def from_nhwc(t, B, H, W, C, ld=None):
    ld = C if ld is None else ld
    return t.view(B, H, W, ld)[..., :C].permute(0, 3, 1, 2).float().cpu()


def pack_bf(w, layout=0):
    N, C = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 4 else 1
    n = lib().nd_conv_bf16_weight_elems(N, C, k)
    assert n > 0
    out = torch.full((n,), float('nan'), dtype=BF, device=DEV)
    wd = w.contiguous().to(DEV)
    _hip.check(lib().nd_repack_conv_weight_bf16(wd.data_ptr(), out.data_ptr(), N, C, k, layout, st()))
    return out


@pytest.mark.parametrize('N,C,k', [(96, 64, 3), (40, 72, 3), (33, 200, 1), (128, 8, 3)])
def test_repack_conv_weight_bf16(N, C, k):
    w = rnd(N, C, k, k, seed=1) if k == 3 else rnd(N, C, seed=1)
    packed = pack_bf(w).cpu()
    taps = k * k
    nt32 = (N + 31) // 32
    nc = ((C + 63) // 64 + 1) & ~1
    assert packed.numel() == (nc + 2) * nt32 * taps * 4 * 512
    p = packed.view(nc + 2, nt32, taps, 4, 64, 8).float()
    wq = q(w).reshape(N, C, taps)
    full = torch.zeros(nt32 * 32, (nc + 2) * 64, taps)
    full[:N, :C] = wq
    # element [c64][nt][tap][ks][lane][j] = w[nt*32 + (lane&31)][c64*64 + ks*16 + (lane>>5)*8 + j][tap]
    lane = torch.arange(64)
    for c64 in (0, nc - 1, nc, nc + 1):
        for ks in range(4):
            for j in (0, 3, 7):
                c = c64 * 64 + ks * 16 + (lane >> 5) * 8 + j
                for nt in (0, nt32 - 1):
                    n = nt * 32 + (lane & 31)
                    for tap in (0, taps - 1):
                        assert torch.equal(p[c64, nt, tap, ks, :, j], full[n, c, tap])
    assert not torch.isnan(p).any()
    # layout 1 (16x16x32 fragments): [c64][n16 tile][tap][ks(2)][lane][8] = w[nt*16 + (lane&15)][c64*64 + ks*32 + (lane>>4)*8 + j][tap]
    p1 = pack_bf(w, 1).cpu().view(nc + 2, 2 * nt32, taps, 2, 64, 8).float()
    for c64 in (0, nc - 1, nc, nc + 1):
        for ks in range(2):
            for j in (0, 5, 7):
                c = c64 * 64 + ks * 32 + (lane >> 4) * 8 + j
                for nt in (0, 2 * nt32 - 1):
                    n = nt * 16 + (lane & 15)
                    for tap in (0, taps - 1):
                        assert torch.equal(p1[c64, nt, tap, ks, :, j], full[n, c, tap])


CONV_CASES = [  # B, Cin, Cout, H, W
    (2, 64, 96, 16, 16), (3, 40, 72, 12, 20), (8, 128, 128, 8, 8), (1, 64, 64, 64, 64), (2, 192, 40, 7, 7),
    (1, 8, 64, 32, 32), (4, 64, 256, 8, 8),
]


def _tol_bf16(ref):
    return 2.0 ** -8 * ref.abs() + 2e-3 * ref.abs().max().clamp(min=1.0)


@pytest.mark.parametrize('B,Cin,Cout,H,W', CONV_CASES)
@pytest.mark.parametrize('ksize', [3, 1])
def test_conv_bf16_all_variants(B, Cin, Cout, H, W, ksize):
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, ksize, ksize, seed=2, scale=0.05)
    b = rnd(Cout, seed=3)
    ref = F.conv2d(q(x).double(), q(w).double(), b.double(), padding=ksize // 2).float()
    xd, bd = nhwc_bf(x), b.to(DEV)
    wds = [pack_bf(w if ksize == 3 else w[:, :, 0, 0], lay) for lay in (0, 1)]
    ran = 0
    for v in list(range(lib().nd_conv_bf16_num_variants())) + [-1]:
        wd = wds[lib().nd_conv_bf16_variant_layout(v)]
        for f32out in (False, True):
            out = torch.full((B * H * W * Cout,), float('nan'), dtype=torch.float32 if f32out else BF, device=DEV)
            rc = lib().nd_conv_bf16_nhwc(xd.data_ptr(), Cin, Cin, None, 0, 0, wd.data_ptr(), bd.data_ptr(), None, 0, None, 0,
                                         out.data_ptr(), Cout, B, H, W, Cout, ksize, _hip.CONV_OUT_F32 if f32out else 0,
                                         v, None, None, 0, st())
            if rc != 0:
                assert v >= 0 and any(m in _hip.last_error() for m in ('no tile variant fits', 'retired variant', 'two-block GEMM form')), \
                    (v, _hip.last_error())
                continue
            ran += 1
            got = from_nhwc(out, B, H, W, Cout)
            err = (got - ref).abs()
            if f32out:      # exact bf16 products, fp32 accumulation: only the summation order differs
                assert err.max().item() < 2e-4 * max(1.0, ref.abs().max().item()), (v, err.max().item())
            else:
                assert (err <= _tol_bf16(ref)).all(), (v, err.max().item())
    assert ran >= 4


@pytest.mark.parametrize('B,C,N,H,W', [(6, 64, 256, 8, 8), (4, 128, 512, 8, 8), (3, 64, 256, 4, 8)])
def test_conv_bf16_several_images_per_block_with_row_bias_and_residual(B, C, N, H, W):
    """8x8 maps put two (or more) images into one 128-pixel block: the compact epilogue then takes the per-image timestep row
    and the output / residual rows per lane (model.py:205,211 on the 8x8 levels of configs[3], [4])."""
    x, w = rnd(B, C, H, W, seed=1), rnd(N, C, 3, 3, seed=2, scale=0.05)
    b, rb, res = rnd(N, seed=3), rnd(B, N, seed=4), rnd(B, N, H, W, seed=5)
    ref = (F.conv2d(q(x).double(), q(w).double(), b.double(), padding=1) + rb.double()[:, :, None, None] + q(res).double()).float()
    xd, wd, bd, rbd, resd = nhwc_bf(x), pack_bf(w), b.to(DEV), rb.to(DEV), nhwc_bf(res)
    outs = []
    for v in (11, 0, 4, 3, -1):
        out = torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV)
        rc = lib().nd_conv_bf16_nhwc(xd.data_ptr(), C, C, None, 0, 0, wd.data_ptr(), bd.data_ptr(), rbd.data_ptr(), N, resd.data_ptr(), N,
                                     out.data_ptr(), N, B, H, W, N, 3, 0, v, None, None, 0, st())
        if rc != 0:
            assert 'no tile variant fits' in _hip.last_error()
            continue
        err = (from_nhwc(out, B, H, W, N) - ref).abs()
        assert (err <= _tol_bf16(ref)).all(), (v, err.max().item())
        outs.append(out)
    assert len(outs) >= 3 and all(torch.equal(outs[0], o) for o in outs[1:])


GEMMQ = 21      # gemm_bf16q_kernel: 128 px x 256 ch, two blocks per CU (nd_gemm_bf16_quad.hip)


@pytest.mark.parametrize('B,C0,C1,N,H,W,res,pad', [
    (2, 128, 0, 256, 16, 16, False, 0), (1, 64, 64, 512, 16, 8, True, 0), (3, 192, 64, 256, 8, 16, True, 16),
    (2, 64, 0, 256, 8, 8, False, 8), (1, 512, 0, 768, 32, 32, True, 0)])
def test_gemm_bf16_two_blocks_per_cu(B, C0, C1, N, H, W, res, pad):
    """Variant 21 of nd_conv_bf16_nhwc (1x1 on a flat pixel list, rows by LDS-DMA, weights global -> VGPR, staged 16-byte
    stores) against float64 on the bf16-rounded operands, and bit for bit against the 3-stage GEMM form (variant 20: same
    order of accumulation); two-source input, residual, strided output rows; what it does not take is refused by name."""
    assert lib().nd_conv_bf16_variant_name(GEMMQ) == b'nd::gemm_bf16q_kernel'
    xa = rnd(B, C0, H, W, seed=1)
    xb = rnd(B, C1, H, W, seed=2) if C1 else None
    w = rnd(N, C0 + C1, seed=3, scale=0.05)
    b = rnd(N, seed=4)
    r = rnd(B, N, H, W, seed=5) if res else None
    xin = q(xa) if xb is None else torch.cat([q(xa), q(xb)], 1)
    ref = F.conv2d(xin.double(), q(w).double()[:, :, None, None], b.double())
    if res:
        ref = ref + q(r).double()
    ref = ref.float()
    xad, xbd = nhwc_bf(xa, C0 + pad), (nhwc_bf(xb, C1 + pad) if C1 else None)
    wd, bd, rd = pack_bf(w), b.to(DEV), (nhwc_bf(r) if res else None)
    ldo = N + pad
    outs = {}
    for v in (GEMMQ, 20):
        out = torch.zeros(B * H * W * ldo, dtype=BF, device=DEV)
        _hip.check(lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0 + pad, _hip.ptr(xbd), C1, C1 + pad if C1 else 0, wd.data_ptr(), bd.data_ptr(),
                                           None, 0, _hip.ptr(rd), N if res else 0, out.data_ptr(), ldo, B, H, W, N, 1, 0, v, None, None, 0, st()),
                   'variant %d' % v)
        outs[v] = out
        err = (from_nhwc(out, B, H, W, N, ldo) - ref).abs()
        assert (err <= _tol_bf16(ref)).all(), (v, err.max().item())
        if pad:
            assert not out.view(B, H, W, ldo)[..., N:].any()
    assert torch.equal(outs[GEMMQ], outs[20])
    # the same launch leaving the per-channel partial statistics of its output behind (nd_conv1x1_bf16_stats_nhwc: the attention
    # block's output projection feeds the next GroupNorm): same output bits, rows = exact sums of the stored values
    rows = lib().nd_conv_bf16_stats_rows(B, H, W, N, GEMMQ)
    if pad == 0:
        assert rows == (H * W // 128 if (H * W) % 128 == 0 else 0)
    if rows > 0 and pad == 0:
        out_s = torch.zeros(B * H * W * N, dtype=BF, device=DEV)
        ps = torch.full((B * rows * 2 * N,), float('nan'), dtype=torch.float32, device=DEV)
        _hip.check(lib().nd_conv1x1_bf16_stats_nhwc(xad.data_ptr(), C0, C0, _hip.ptr(xbd), C1, C1, wd.data_ptr(), bd.data_ptr(), None, 0,
                                                    _hip.ptr(rd), N if res else 0, out_s.data_ptr(), N, B, H, W, N, 0, GEMMQ, None, None, 0,
                                                    ps.data_ptr(), st()))
        assert torch.equal(out_s, outs[GEMMQ]) and not torch.isnan(ps).any()
        o = out_s.view(B, H * W, N).double()
        pp = ps.view(B, rows, 2, N).double().sum(1)
        assert (pp[:, 0] - o.sum(1)).abs().max().item() <= 1e-5 * max(1.0, o.abs().sum(1).max().item())
        assert (pp[:, 1] - (o * o).sum(1)).abs().max().item() <= 1e-5 * (o * o).sum(1).max().item()
        # a variant that cannot is refused by name
        assert lib().nd_conv1x1_bf16_stats_nhwc(xad.data_ptr(), C0, C0, _hip.ptr(xbd), C1, C1, wd.data_ptr(), bd.data_ptr(), None, 0,
                                                _hip.ptr(rd), N if res else 0, out_s.data_ptr(), N, B, H, W, N, 0, 20, None, None, 0,
                                                ps.data_ptr(), st()) != 0
    # refused, with the reason: fp32 output, SiLU, N not a multiple of 256, M not a multiple of 128, a per-image bias row
    out32 = torch.zeros(B * H * W * ldo, dtype=torch.float32, device=DEV)
    rc = lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0 + pad, _hip.ptr(xbd), C1, C1 + pad if C1 else 0, wd.data_ptr(), bd.data_ptr(), None, 0,
                                 None, 0, out32.data_ptr(), ldo, B, H, W, N, 1, _hip.CONV_OUT_F32, GEMMQ, None, None, 0, st())
    assert rc != 0 and 'two-block GEMM form' in _hip.last_error()
    rc = lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0 + pad, _hip.ptr(xbd), C1, C1 + pad if C1 else 0, wd.data_ptr(), bd.data_ptr(), None, 0,
                                 None, 0, outs[20].data_ptr(), ldo, B, H, W, N - 32, 1, 0, GEMMQ, None, None, 0, st())
    assert rc != 0 and 'two-block GEMM form' in _hip.last_error()


def test_gemm_bf16_two_blocks_per_cu_run_to_run():
    """gemm_bf16q_kernel issues its operand loads as inline ISA with hand-counted waits: a register that the compiler re-used
    while a load into it was still in flight would show as run-to-run differences (the failure conv_wino4_kernel had in this
    round before its fragments were tied to the final wait).  Short K, several n blocks, residual, many launches back to
    back on a busy chip: every output must equal the 3-stage form's, bit for bit, every time."""
    for (B, C, N, H, W) in ((8, 64, 768, 16, 16), (4, 128, 256, 32, 32), (2, 512, 1536, 32, 32)):
        x, w, b, r = rnd(B, C, H, W, seed=11), rnd(N, C, seed=12, scale=0.05), rnd(N, seed=13), rnd(B, N, H, W, seed=14)
        xd, wd, bd, rd = nhwc_bf(x), pack_bf(w), b.to(DEV), nhwc_bf(r)
        ref = torch.zeros(B * H * W * N, dtype=BF, device=DEV)
        _hip.check(lib().nd_conv_bf16_nhwc(xd.data_ptr(), C, C, None, 0, 0, wd.data_ptr(), bd.data_ptr(), None, 0, rd.data_ptr(), N,
                                           ref.data_ptr(), N, B, H, W, N, 1, 0, 20, None, None, 0, st()))
        outs = [torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV) for _ in range(24)]
        for o in outs:
            _hip.check(lib().nd_conv_bf16_nhwc(xd.data_ptr(), C, C, None, 0, 0, wd.data_ptr(), bd.data_ptr(), None, 0, rd.data_ptr(), N,
                                               o.data_ptr(), N, B, H, W, N, 1, 0, GEMMQ, None, None, 0, st()))
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, ref), (B, C, N, i)


def test_conv_bf16_fused_options():
    """Two-source input (torch.cat), per-image bias, residual, nearest-2x input / residual, SiLU -- the options of the
    fp32 kernel (model.py:474, :205, :211, :77-79)."""
    B, C0, C1, N, H, W = 2, 64, 40, 96, 16, 16
    xa, xb = rnd(B, C0, H, W, seed=1), rnd(B, C1, H, W, seed=2)
    w = rnd(N, C0 + C1, 3, 3, seed=3, scale=0.05)
    b, rb, res = rnd(N, seed=4), rnd(B, N, seed=5), rnd(B, N, H, W, seed=6)
    xad, xbd, wd = nhwc_bf(xa), nhwc_bf(xb), pack_bf(w)
    bd, rbd, resd = b.to(DEV), rb.to(DEV), nhwc_bf(res)
    ref = F.conv2d(torch.cat([q(xa), q(xb)], 1).double(), q(w).double(), b.double(), padding=1) + rb.double()[:, :, None, None] \
        + q(res).double()
    for v in (0, 4, 7, -1):
        out = torch.empty(B * H * W * N, dtype=BF, device=DEV)
        _hip.check(lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0, xbd.data_ptr(), C1, C1, wd.data_ptr(), bd.data_ptr(),
                                           rbd.data_ptr(), N, resd.data_ptr(), N, out.data_ptr(), N, B, H, W, N, 3, 0, v, None, None, 0, st()))
        err = (from_nhwc(out, B, H, W, N) - ref.float()).abs()
        assert (err <= _tol_bf16(ref.float())).all(), (v, err.max().item())
    # SiLU on the output; 1x1 with a strided output row (ldo > N) and fp32 output
    w1 = rnd(N, C0, seed=7, scale=0.05)
    wd1 = pack_bf(w1)
    ref1 = F.silu(F.conv2d(q(xa).double(), q(w1).double()[:, :, None, None], b.double())).float()
    out = torch.zeros(B * H * W * (N + 8), dtype=torch.float32, device=DEV)
    _hip.check(lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0, None, 0, 0, wd1.data_ptr(), bd.data_ptr(), None, 0, None, 0,
                                       out.data_ptr(), N + 8, B, H, W, N, 1, _hip.CONV_SILU_OUT | _hip.CONV_OUT_F32, -1, None, None, 0, st()))
    assert (from_nhwc(out, B, H, W, N, N + 8) - ref1).abs().max().item() < 2e-4
    assert not out.view(B, H, W, N + 8)[..., N:].any()
    # nearest-2x upsampled input and residual
    xs, rs = rnd(B, C0, H // 2, W // 2, seed=8), rnd(B, N, H // 2, W // 2, seed=9)
    w3 = rnd(N, C0, 3, 3, seed=10, scale=0.05)
    ref_up = F.conv2d(F.interpolate(q(xs), scale_factor=2.0, mode='nearest').double(), q(w3).double(), b.double(), padding=1)
    xsd, rsd, wd3 = nhwc_bf(xs), nhwc_bf(rs), pack_bf(w3)
    out = torch.empty(B * H * W * N, dtype=BF, device=DEV)
    _hip.check(lib().nd_conv_bf16_nhwc(xsd.data_ptr(), C0, C0, None, 0, 0, wd3.data_ptr(), bd.data_ptr(), None, 0, None, 0,
                                       out.data_ptr(), N, B, H, W, N, 3, _hip.CONV_IN_UP2X, -1, None, None, 0, st()))
    err = (from_nhwc(out, B, H, W, N) - ref_up.float()).abs()
    assert (err <= _tol_bf16(ref_up.float())).all()
    ref_r = F.conv2d(q(xa).double(), q(w3).double(), b.double(), padding=1) + F.interpolate(q(rs), scale_factor=2.0, mode='nearest').double()
    _hip.check(lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0, None, 0, 0, wd3.data_ptr(), bd.data_ptr(), None, 0,
                                       rsd.data_ptr(), N, out.data_ptr(), N, B, H, W, N, 3, _hip.CONV_RES_UP2X, -1, None, None, 0, st()))
    err = (from_nhwc(out, B, H, W, N) - ref_r.float()).abs()
    assert (err <= _tol_bf16(ref_r.float())).all()


@pytest.mark.parametrize('ksize,silu', [(3, True), (1, False), (3, False)])
def test_conv_bf16_fused_groupnorm(ksize, silu):
    """GroupNorm(+AdaGN)(+SiLU) of a two-source input applied by the conv's loader (coefficients from nd_groupnorm_coeffs)
    = the explicit bf16 apply pass followed by the plain conv, up to the one bf16 rounding of the normalised tensor that
    the fused form shares; zero padding must stay zero AFTER normalisation (model.py:190-194)."""
    B, C0, C1, N, H, W = 2, 64, 32, 96, 16, 16
    C = C0 + C1
    xa, xb = rnd(B, C0, H, W, seed=1) * 2 + 0.5, rnd(B, C1, H, W, seed=2)
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.5 + 0.1 * rnd(C, seed=4)       # beta != 0: padding would show up
    scale, shift = 0.3 * rnd(B, C, seed=5), 0.3 * rnd(B, C, seed=6)
    w = rnd(N, C, ksize, ksize, seed=7, scale=0.05)
    b = rnd(N, seed=8)
    x = q(torch.cat([xa, xb], 1))
    h = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5) * (1 + scale.double()[:, :, None, None]) \
        + shift.double()[:, :, None, None]
    if silu:
        h = F.silu(h)
    ref = F.conv2d(q(h.float()).double(), q(w).double(), b.double(), padding=ksize // 2).float()
    xad, xbd, bd = nhwc_bf(xa), nhwc_bf(xb), b.to(DEV)
    wd = pack_bf(w if ksize == 3 else w[:, :, 0, 0])
    stats, nb = gn_stats(xad, C0, xbd, C1, B, H * W, _hip.DT_BF16)
    gd, btd, scd, shd = gamma.to(DEV), beta.to(DEV), scale.to(DEV), shift.to(DEV)
    cA, cB = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV)
    _hip.check(lib().nd_groupnorm_coeffs(stats.data_ptr(), nb, gd.data_ptr(), btd.data_ptr(), scd.data_ptr(), shd.data_ptr(), C,
                                         cA.data_ptr(), cB.data_ptr(), C, B, C, H * W, 32, 1e-5, st()))
    flags = _hip.CONV_GN_SILU if silu else 0
    ran = 0
    wds = [wd, pack_bf(w if ksize == 3 else w[:, :, 0, 0], 1)]
    for v in range(lib().nd_conv_bf16_num_variants()):
        out = torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV)
        wd = wds[lib().nd_conv_bf16_variant_layout(v)]
        rc = lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0, xbd.data_ptr(), C1, C1, wd.data_ptr(), bd.data_ptr(), None, 0,
                                     None, 0, out.data_ptr(), N, B, H, W, N, ksize, flags, v, cA.data_ptr(), cB.data_ptr(), C,
                                     st())
        if rc != 0:
            continue            # tile shape does not fit / variant cannot fuse
        ran += 1
        err = (from_nhwc(out, B, H, W, N) - ref).abs()
        # the normalised activations are rounded to bf16 once (as in the unfused path); fp32 vs fp64 coefficient
        # arithmetic can move a value across a rounding boundary, hence a slightly wider bound than the plain conv
        assert (err <= 2.0 ** -7 * ref.abs() + 6e-3 * ref.abs().max().clamp(min=1.0)).all(), (v, err.max().item())
    assert ran >= 4
    # several images per block are refused (callers materialise the normalised tensor instead)
    rc = lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0, None, 0, 0, wd.data_ptr(), bd.data_ptr(), None, 0, None, 0,
                                 out.data_ptr(), N, B * 4, 8, 8, N, ksize, flags, 0, cA.data_ptr(), cB.data_ptr(), C, st())
    assert rc == -1


@pytest.mark.parametrize('B,C0,C1,N,H,W,opts', [
    (2, 64, 0, 96, 16, 16, ''), (2, 64, 32, 72, 32, 32, 'rowbias,residual'), (3, 40, 0, 256, 20, 12, 'residual'),
    (1, 64, 0, 64, 64, 64, 'gn'), (2, 64, 0, 96, 16, 16, 'res_up'), (2, 32, 0, 48, 32, 32, 'up'),
    # whole 256-channel n blocks on full tiles: the compact staged epilogue (statistics formed while draining the LDS region)
    (2, 64, 0, 256, 32, 32, ''), (1, 128, 64, 512, 16, 16, 'rowbias,residual'), (2, 64, 0, 256, 16, 16, 'gn'),
    (1, 64, 0, 256, 32, 32, 'res_up'),
    # <= 16 input channels (the UNet's first convolution): variant 11 skips the three empty k-steps of every tap
    (2, 8, 0, 256, 32, 32, ''), (1, 8, 8, 256, 16, 16, 'rowbias')])
def test_conv3x3_bf16_epilogue_statistics(B, C0, C1, N, H, W, opts):
    """nd_conv3x3_bf16_stats_nhwc: bit-identical output to nd_conv_bf16_nhwc with the same variant, plus partial per-channel
    sums / sums of squares of the bf16 values it stored; nd_groupnorm_stats_from_partials folds them into what the
    statistics pass over the output computes (same numbers up to fp32 partial sums of <= 256 values), also for a
    two-source consumer; repeated launches give the same bits (no atomics)."""
    C = C0 + C1
    up = 'up' in opts.split(',')
    Hs, Ws = (H // 2, W // 2) if up else (H, W)
    xa = rnd(B, C0, Hs, Ws, seed=1)
    xb = rnd(B, C1, Hs, Ws, seed=2) if C1 else None
    w = rnd(N, C, 3, 3, seed=3, scale=0.05)
    b, rb = rnd(N, seed=4), rnd(B, N, seed=5)
    res_up = 'res_up' in opts
    res = rnd(B, N, H // 2 if res_up else H, W // 2 if res_up else W, seed=6)
    xad, xbd = nhwc_bf(xa), (nhwc_bf(xb) if C1 else None)
    wd, bd, rbd, resd = pack_bf(w), b.to(DEV), rb.to(DEV), nhwc_bf(res)
    use_rb, use_res = 'rowbias' in opts, ('residual' in opts or res_up)
    flags = (_hip.CONV_IN_UP2X if up else 0) | (_hip.CONV_RES_UP2X if res_up else 0)
    gn = [None, None, 0]
    if 'gn' in opts:
        cA, cB = (1 + 0.2 * rnd(B, C, seed=7)).to(DEV).contiguous(), (0.1 * rnd(B, C, seed=8)).to(DEV).contiguous()
        gn = [cA.data_ptr(), cB.data_ptr(), C]
        flags |= _hip.CONV_GN_SILU
    head = [xad.data_ptr(), C0, C0, None if not C1 else xbd.data_ptr(), C1, C1, wd.data_ptr(), bd.data_ptr(),
            rbd.data_ptr() if use_rb else None, N if use_rb else 0, resd.data_ptr() if use_res else None, N if use_res else 0]
    ran = 0
    for v in range(lib().nd_conv_bf16_num_variants()):
        rows = lib().nd_conv_bf16_stats_rows(B, H, W, N, v)
        out0 = torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV)
        rc0 = lib().nd_conv_bf16_nhwc(*head, out0.data_ptr(), N, B, H, W, N, 3, flags, v, *gn, st())
        if rows <= 0:
            dummy = torch.empty(16, device=DEV)
            assert lib().nd_conv3x3_bf16_stats_nhwc(*head, out0.data_ptr(), N, B, H, W, N, flags, v, *gn, dummy.data_ptr(), st()) != 0
            continue
        if rc0 != 0:
            continue
        ran += 1
        out1 = torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV)
        ps = torch.full((B * rows * 2 * N,), float('nan'), dtype=torch.float32, device=DEV)
        _hip.check(lib().nd_conv3x3_bf16_stats_nhwc(*head, out1.data_ptr(), N, B, H, W, N, flags, v, *gn, ps.data_ptr(), st()))
        assert torch.equal(out0.view(torch.int16), out1.view(torch.int16)), v
        assert not torch.isnan(ps).any(), v
        o = out1.view(B, H * W, N).double()
        pp = ps.view(B, rows, 2, N).double().sum(1)
        assert (pp[:, 0] - o.sum(1)).abs().max().item() <= 1e-5 * max(1.0, o.abs().sum(1).max().item()), v
        assert (pp[:, 1] - (o * o).sum(1)).abs().max().item() <= 1e-5 * (o * o).sum(1).max().item(), v
        if N % 32 == 0:
            a = torch.empty(B * 64, dtype=torch.float64, device=DEV)
            _hip.check(lib().nd_groupnorm_stats_from_partials(ps.data_ptr(), N, rows, None, 0, 0, a.data_ptr(), B, 32, st()))
            ref = torch.stack([o.view(B, H * W, 32, N // 32).sum((1, 3)), (o * o).view(B, H * W, 32, N // 32).sum((1, 3))], -1)
            assert (a.view(B, 32, 2) - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), v
            # consumer reading cat([out, out]) (the up path's skip concatenation; groups straddle the seam when N/16 is odd)
            a2 = torch.empty(B * 64, dtype=torch.float64, device=DEV)
            _hip.check(lib().nd_groupnorm_stats_from_partials(ps.data_ptr(), N, rows, ps.data_ptr(), N, rows, a2.data_ptr(), B, 32, st()))
            o2 = torch.cat([o, o], 2)
            ref2 = torch.stack([o2.view(B, H * W, 32, N // 16).sum((1, 3)), (o2 * o2).view(B, H * W, 32, N // 16).sum((1, 3))], -1)
            assert (a2.view(B, 32, 2) - ref2).abs().max().item() <= 1e-5 * ref2.abs().max().item(), v
        ps2 = torch.empty_like(ps)
        _hip.check(lib().nd_conv3x3_bf16_stats_nhwc(*head, out1.data_ptr(), N, B, H, W, N, flags, v, *gn, ps2.data_ptr(), st()))
        assert torch.equal(ps, ps2), v
    assert ran >= 3


@pytest.mark.parametrize('base', ['adagn_updown', 'plain_convres_legacy'])
def test_bf16_forward_with_epilogue_statistics_matches_statistics_pass(monkeypatch, base):
    """The plan builder's choice (ND_BF16_EPILOGUE_STATS: 1 = where measured faster, 2 = wherever possible) only moves
    where a GroupNorm's sums are formed: the forward with conv-epilogue statistics equals the forward with the statistics
    pass to within a few bf16 roundings, and the epilogue form is actually taken on a model whose convs are big enough to
    be tuned (AdaGN + resblock up/down, and plain GN with the embedding added in the conv epilogue + conv resampling)."""
    from nicediffusion import _engine
    cfg = dict(TINY_CFGS[base], resolution=32, model_channels=64, attention_resolutions=(16,))
    B = 8
    x = rnd(B, 3, 32, 32, seed=3).to(DEV)
    t = torch.tensor([5, 100, 300, 999, 0, 1, 2, 3], device=DEV)
    y = (torch.arange(B, device=DEV) % 10) if cfg.get('num_classes') else None
    outs = {}
    monkeypatch.setenv('ND_GN_FUSED_MAX', '0')        # (tensors this small would take the one-launch GroupNorm: not under test here)
    for mode in ('2', '0'):
        monkeypatch.setenv('ND_BF16_EPILOGUE_STATS', mode)
        _engine._TUNED.clear()
        m = build(cfg)
        outs[mode] = m(x, t, y=y).float().cpu()
        plan = next(iter(m._plans.values()))
        used = sum(1 for f, _, _ in plan.ops if f.__name__ == 'nd_conv3x3_bf16_stats_nhwc')
        folds = sum(1 for f, _, _ in plan.ops if f.__name__ in ('nd_groupnorm_stats_from_partials', 'nd_groupnorm_coeffs_from_partials'))
        if mode == '0':
            assert used == 0 and folds == 0
        else:
            print('epilogue statistics on %d convs, %d folds' % (used, folds))
            assert used > 0 and folds > 0
    _engine._TUNED.clear()
    assert torch.isfinite(outs['2']).all()
    rms, mx = _errs(outs['2'], outs['0'])
    print('epilogue statistics vs statistics pass: rms %.2e max %.2e' % (rms, mx))
    assert rms < 1e-2 and mx < 4e-2


@pytest.mark.parametrize('B,C0,C1,N,H,W,ksize,opts', [
    (4, 512, 0, 256, 8, 8, 3, 'residual'), (2, 320, 192, 128, 8, 8, 3, 'rowbias'), (3, 256, 128, 96, 16, 16, 1, 'residual'),
    (2, 1024, 0, 64, 8, 8, 1, 'silu'), (1, 200, 56, 72, 8, 8, 3, '')])
def test_conv_bf16_split_k(B, C0, C1, N, H, W, ksize, opts):
    """nd_conv_bf16_splitk_nhwc (block rows over ranges of input-channel chunks, fp32 partials, ordered reduce with bias /
    per-image bias / residual / SiLU) against the float64 statement of the op and against the one-pass kernel: the two differ
    only by the association of the fp32 sums (<= 1 bf16 ulp after rounding); two-source inputs whose seam falls inside a
    split, channel tails, repeated launches give the same bits."""
    C = C0 + C1
    xa = rnd(B, C0, H, W, seed=1)
    xb = rnd(B, C1, H, W, seed=2) if C1 else None
    w = rnd(N, C, ksize, ksize, seed=3, scale=0.03)
    b, rb, res = rnd(N, seed=4), rnd(B, N, seed=5), rnd(B, N, H, W, seed=6)
    x = q(xa) if not C1 else torch.cat([q(xa), q(xb)], 1)
    ref = F.conv2d(x.double(), q(w).double(), b.double(), padding=ksize // 2)
    if 'rowbias' in opts:
        ref = ref + rb.double()[:, :, None, None]
    if 'residual' in opts:
        ref = ref + q(res).double()
    if 'silu' in opts:
        ref = F.silu(ref)
    ref = ref.float()
    xad, xbd = nhwc_bf(xa), (nhwc_bf(xb) if C1 else None)
    bd, rbd, resd = b.to(DEV), rb.to(DEV), nhwc_bf(res)
    wds = [pack_bf(w if ksize == 3 else w[:, :, 0, 0], lay) for lay in (0, 1)]
    flags = _hip.CONV_SILU_OUT if 'silu' in opts else 0
    head = [xad.data_ptr(), C0, C0, None if not C1 else xbd.data_ptr(), C1, C1, None, bd.data_ptr(),
            rbd.data_ptr() if 'rowbias' in opts else None, N if 'rowbias' in opts else 0,
            resd.data_ptr() if 'residual' in opts else None, N if 'residual' in opts else 0]
    ran = 0
    for v in range(lib().nd_conv_bf16_num_variants()):
        head[6] = wds[lib().nd_conv_bf16_variant_layout(v)].data_ptr()
        plain = torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV)
        if lib().nd_conv_bf16_nhwc(*head, plain.data_ptr(), N, B, H, W, N, ksize, flags, v, None, None, 0, st()) != 0:
            continue
        for S in (2, 3, 4):
            ws = torch.full((S * B * H * W * N,), float('nan'), dtype=torch.float32, device=DEV)
            out = torch.full((B * H * W * N,), float('nan'), dtype=BF, device=DEV)
            rc = lib().nd_conv_bf16_splitk_nhwc(*head, out.data_ptr(), N, B, H, W, N, ksize, flags, v, S, ws.data_ptr(), st())
            if rc != 0:
                assert 'split-K' in _hip.last_error(), _hip.last_error()
                continue
            ran += 1
            got = from_nhwc(out, B, H, W, N)
            assert torch.isfinite(got).all(), (v, S)
            err = (got - ref).abs()
            assert (err <= _tol_bf16(ref)).all(), (v, S, err.max().item())
            d = (got - from_nhwc(plain, B, H, W, N)).abs()
            assert (d <= 2.0 ** -7 * ref.abs() + 1e-3).all(), (v, S, d.max().item())
            out2 = torch.empty_like(out)
            _hip.check(lib().nd_conv_bf16_splitk_nhwc(*head, out2.data_ptr(), N, B, H, W, N, ksize, flags, v, S, ws.data_ptr(), st()))
            assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), (v, S)
    assert ran >= 4
    # refused: fused GroupNorm is not an argument; upsampled reads and fp32 output are rejected
    out = torch.empty(B * H * W * N, dtype=BF, device=DEV)
    ws = torch.empty(2 * B * H * W * N, dtype=torch.float32, device=DEV)
    head[6] = wds[0].data_ptr()
    assert lib().nd_conv_bf16_splitk_nhwc(*head, out.data_ptr(), N, B, H, W, N, ksize, _hip.CONV_OUT_F32, 7, 2, ws.data_ptr(), st()) != 0
    assert lib().nd_conv_bf16_splitk_nhwc(*head, out.data_ptr(), N, B, H, W, N, ksize, flags, -1, 2, ws.data_ptr(), st()) != 0


def test_bf16_forward_with_split_k_matches_one_pass(monkeypatch):
    """ND_BF16_SPLITK=2 (split-K wherever a split form exists) against ND_BF16_SPLITK=0 on a model with 512 channels at
    8x8: the forwards agree to within a few bf16 roundings and the split form is actually in the plan."""
    from nicediffusion import _engine
    cfg = dict(TINY_CFGS['adagn_updown'], resolution=16, model_channels=256, attention_resolutions=(8,), num_head_channels=64)
    B = 8
    x = rnd(B, 3, 16, 16, seed=3).to(DEV)
    t = torch.tensor([5, 100, 300, 999, 0, 1, 2, 3], device=DEV)
    y = torch.arange(B, device=DEV) % 10
    outs = {}
    for mode in ('2', '0'):
        monkeypatch.setenv('ND_BF16_SPLITK', mode)
        _engine._TUNED.clear()
        m = build(cfg)
        outs[mode] = m(x, t, y=y).float().cpu()
        plan = next(iter(m._plans.values()))
        used = sum(1 for f, _, _ in plan.ops if f.__name__ == 'nd_conv_bf16_splitk_nhwc')
        assert (used > 0) == (mode == '2'), (mode, used)
    _engine._TUNED.clear()
    assert torch.isfinite(outs['2']).all()
    rms, mx = _errs(outs['2'], outs['0'])
    print('split-K vs one pass: rms %.2e max %.2e' % (rms, mx))
    assert rms < 1e-2 and mx < 4e-2


def test_conv_bf16_long_k_full_size_layer():
    """A full-size layer of the 128x128 preset (two-source 512+256 -> 256 at 64x64, B=2): long contraction (K = 6912),
    every variant that fits gives the same result as the cost model's pick to within output rounding."""
    B, C0, C1, N, H, W = 2, 512, 256, 256, 64, 64
    xa, xb = rnd(B, C0, H, W, seed=1), rnd(B, C1, H, W, seed=2)
    w = rnd(N, C0 + C1, 3, 3, seed=3, scale=0.02)
    b = rnd(N, seed=4)
    xad, xbd, wd, bd = nhwc_bf(xa), nhwc_bf(xb), pack_bf(w), b.to(DEV)
    ref = F.conv2d(torch.cat([q(xa), q(xb)], 1), q(w), b, padding=1)       # fp32 CPU conv on the same bf16 operands
    outs = []
    wds = [wd, pack_bf(w, 1)]
    for v in range(lib().nd_conv_bf16_num_variants()):
        out = torch.empty(B * H * W * N, dtype=torch.float32, device=DEV)
        wd = wds[lib().nd_conv_bf16_variant_layout(v)]
        rc = lib().nd_conv_bf16_nhwc(xad.data_ptr(), C0, C0, xbd.data_ptr(), C1, C1, wd.data_ptr(), bd.data_ptr(), None, 0,
                                     None, 0, out.data_ptr(), N, B, H, W, N, 3, _hip.CONV_OUT_F32, v, None, None, 0, st())
        if rc == 0:
            outs.append((v, from_nhwc(out, B, H, W, N)))
    assert len(outs) >= 6
    for v, o in outs:
        assert (o - ref).abs().max().item() < 1e-3 * ref.abs().max().item(), v


def gn_stats(xd, C0, x1d, C1, B, HW, dtype):
    nb = lib().nd_groupnorm_stats_blocks(B, HW, C0 + C1, dtype)
    part = torch.full((B * nb * 64,), float('nan'), dtype=torch.float64, device=DEV)
    _hip.check(lib().nd_groupnorm_stats_nhwc(xd.data_ptr(), C0, C0, None if x1d is None else x1d.data_ptr(), C1, C1, None, 0,
                                             part.data_ptr(), B, HW, 32, dtype, st()))
    return part, nb


@pytest.mark.parametrize('B,C0,C1,H,W', [(2, 64, 0, 16, 16), (3, 192, 64, 8, 8), (2, 256, 0, 64, 64), (1, 1024, 1024, 8, 8)])
@pytest.mark.parametrize('mode', ['silu', 'adagn', 'pool'])
def test_groupnorm_bf16(B, C0, C1, H, W, mode):
    C = C0 + C1
    xa = rnd(B, C0, H, W, seed=1) * 2 + 0.5
    xb = rnd(B, C1, H, W, seed=2) if C1 else None
    x = q(torch.cat([xa, xb], 1) if C1 else xa)
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    scale, shift = 0.3 * rnd(B, C, seed=5), 0.3 * rnd(B, C, seed=6)
    ref = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)
    if mode == 'adagn':
        ref = ref * (1 + scale.double()[:, :, None, None]) + shift.double()[:, :, None, None]
    ref = F.silu(ref)
    if mode == 'pool':
        ref = F.avg_pool2d(ref, 2, 2)
    xad, xbd = nhwc_bf(xa), (nhwc_bf(xb) if C1 else None)
    stats, nb = gn_stats(xad, C0, xbd, C1, B, H * W, _hip.DT_BF16)
    assert torch.equal(stats, gn_stats(xad, C0, xbd, C1, B, H * W, _hip.DT_BF16)[0])          # reproducible bits
    s = stats.cpu().view(B, nb, 32, 2).sum(1)
    xg = x.double().view(B, 32, -1)
    assert torch.allclose(s[..., 0], xg.sum(-1), rtol=1e-12, atol=1e-9)
    assert torch.allclose(s[..., 1], (xg * xg).sum(-1), rtol=1e-12, atol=1e-9)
    Ho, Wo = (H // 2, W // 2) if mode == 'pool' else (H, W)
    out = torch.empty(B * Ho * Wo * C, dtype=BF, device=DEV)
    p = lambda t: None if t is None else t.data_ptr()
    sc, sh = (scale.to(DEV), shift.to(DEV)) if mode == 'adagn' else (None, None)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    _hip.check(lib().nd_groupnorm_apply_nhwc(xad.data_ptr(), C0, C0, p(xbd), C1, C1, None, 0, stats.data_ptr(), nb, gd.data_ptr(),
                                             bd.data_ptr(), p(sc), p(sh), C, out.data_ptr(), C, B, H, W, 32, 1e-5,
                                             _hip.GN_SILU | (_hip.GN_POOL2 if mode == 'pool' else 0), _hip.DT_BF16, st()))
    got = from_nhwc(out, B, Ho, Wo, C)
    assert ((got - ref.float()).abs() <= 2.0 ** -8 * ref.float().abs() + 1e-4).all()


@pytest.mark.parametrize('B,T,nh,hd,split', [(2, 64, 2, 64, True), (1, 1024, 4, 64, True), (2, 256, 3, 64, False),
                                             (3, 196, 2, 32, True), (2, 49, 4, 64, True), (1, 1024, 4, 128, True),
                                             (1, 256, 4, 192, True), (2, 64, 4, 256, True), (1, 64, 2, 16, False)])
def test_attention_bf16(B, T, nh, hd, split):
    C = nh * hd
    qkv = rnd(B, T, 3 * C, seed=1)
    qkv[0, T // 2, :] *= 3.0            # a spiky token: exercises the online-softmax rescale
    qq = q(qkv)
    if split:
        Q, K, V = qq.view(B, T, 3, nh, hd).permute(2, 0, 3, 1, 4)
        offs = (0, C, 2 * C, hd)
    else:
        Q, K, V = qq.view(B, T, nh, 3, hd).permute(3, 0, 2, 1, 4)
        offs = (0, hd, 2 * hd, 3 * hd)
    scale = hd ** -0.5
    wgt = torch.softmax(Q.double() @ K.double().transpose(-1, -2) * scale, -1)
    ref = (wgt @ V.double()).permute(0, 2, 1, 3).reshape(B, T, C).float()
    qd = qkv.to(BF).contiguous().to(DEV)
    out = torch.full((B * T * C,), float('nan'), dtype=BF, device=DEV)
    _hip.check(lib().nd_attention_bf16_nhwc(qd.data_ptr(), 3 * C, out.data_ptr(), C, B, T, nh, hd, offs[0], offs[1], offs[2],
                                            offs[3], scale, st()))
    got = out.view(B, T, C).float().cpu()
    err = (got - ref).abs().max().item()
    # probabilities and the output are rounded to bf16 (2^-9 relative each); the spiky token makes outputs of magnitude 3-4
    assert err < 2.0 ** -7 * max(1.0, ref.abs().max().item()) + 5e-3, (err, ref.abs().max().item())
    assert ((got - ref) ** 2).mean().sqrt().item() < 3e-3


def test_avgpool_and_input_cast_bf16():
    B, C, H, W = 2, 64, 8, 12
    x = rnd(B, C, H, W, seed=1)
    xd = nhwc_bf(x)
    dn = torch.empty(B * (H // 2) * (W // 2) * C, dtype=BF, device=DEV)
    _hip.check(lib().nd_avgpool2x_nhwc(xd.data_ptr(), C, dn.data_ptr(), C, B, H, W, C, _hip.DT_BF16, st()))
    ref = F.avg_pool2d(q(x).double(), 2, 2).float()
    assert torch.equal(from_nhwc(dn, B, H // 2, W // 2, C), q(ref))
    # bf16 tensors through the pure-movement kernels as fp32 words of C/2 channels
    up = torch.empty(B * 4 * H * W * C, dtype=BF, device=DEV)
    _hip.check(lib().nd_upsample2x_nhwc(xd.data_ptr(), C // 2, up.data_ptr(), C // 2, B, H, W, C // 2, st()))
    assert torch.equal(from_nhwc(up, B, 2 * H, 2 * W, C), F.interpolate(q(x), scale_factor=2.0, mode='nearest'))
    # fp32 NHWC4 image -> bf16 NHWC8 (3 channels kept, 5 zero)
    img = torch.randn(5 * 7, 4, device=DEV)
    o = torch.full((5 * 7 * 8,), float('nan'), dtype=BF, device=DEV)
    _hip.check(lib().nd_f32_to_bf16_rows(img.data_ptr(), 4, o.data_ptr(), 8, 3, 5 * 7, st()))
    ov = o.view(35, 8)
    assert torch.equal(ov[:, :3], img[:, :3].to(BF)) and not ov[:, 3:].any()


# ---------------------------------------------------------------------------------------------------- model level
def build(cfg, seed=1234, dtype='bf16', **kw):
    m = DiffusionModel(**cfg)
    m.load_state_dict(UO.synth_state_dict(cfg, seed=seed, **kw), strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = dtype
    return m


def _errs(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    d = got - ref
    return float(np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean())), float(np.abs(d).max() / np.abs(ref).max())


@pytest.mark.parametrize('name', sorted(n for n in TINY_CFGS if n != 'odd_sizes') + ['odd_sizes'])
def test_tiny_forward_bf16_vs_fp32_reference_golden(golden_dir, name):
    """bf16 forward vs the REFERENCE's fp32 output (tests/golden) -- every code path of the four tiny configurations,
    with the per-block intermediates compared too so that an error is attributed to a block."""
    g = np.load(os.path.join(golden_dir, 'fwd_{}.npz'.format(name)))
    m = build(TINY_CFGS[name])
    y = torch.from_numpy(g['y']).to(DEV) if 'y' in g.files else None
    out = m(torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['t']).to(DEV), y).cpu().numpy()
    rms, mx = _errs(out, g['out'])
    print('bf16 forward {}: rel rms {:.2e}, max/absmax {:.2e}'.format(name, rms, mx))
    assert rms < TINY_FWD_BF16_BOUND[name][0] and mx < TINY_FWD_BF16_BOUND[name][1], (rms, mx)
    B = g['x'].shape[0]
    plan = m._plan(B)
    assert plan.bf16
    got = plan.run_with_taps()
    worst = {}
    for k in (k[4:] for k in g.files if k.startswith('tap/')):
        worst[k] = _errs(got[k].cpu().numpy(), g['tap/' + k])[0]
    print('bf16 per-block worst rel rms {}: {:.2e}'.format(name, max(worst.values())))
    bad = {k: v for k, v in worst.items() if not v < TAP_BF16_BOUND}
    assert not bad, bad


@pytest.mark.parametrize('pname,B', [('EMNIST', 4), ('OPENAI_64', 2), ('OPENAI_128', 1), ('OPENAI_256', 1)])
def test_preset_forward_bf16_vs_fp32(pname, B):
    """All four presets: bf16 forward against the fp32 HIP forward of the same weights (itself pinned to the reference
    by the fp32 tests).  Tolerance: relative rms 3.5e-2, max 8e-2 of the output's max magnitude."""
    margs = dict(getattr(DA, pname + '_MODEL_ARGS'))
    m = build(margs, dtype='fp32')
    R, C = margs['resolution'], margs['in_channels']
    torch.manual_seed(0)
    x = torch.randn(B, C, R, R).to(DEV)
    t = torch.tensor([17, 250, 600, 990][:B]).to(DEV)
    y = ((torch.arange(B) * 37) % margs['num_classes']).to(DEV) if margs.get('num_classes') else None
    ref = m(x, t, y).cpu().numpy()
    m.compute_dtype = 'bf16'
    got = m(x, t, y).cpu().numpy()
    assert np.isfinite(got).all()
    rms, mx = _errs(got, ref)
    print('bf16 forward preset {}: rel rms {:.2e}, max/absmax {:.2e}'.format(pname, rms, mx))
    assert rms < PRESET_FWD_BF16_BOUND[pname][0] and mx < PRESET_FWD_BF16_BOUND[pname][1], (rms, mx)
    again = m(x, t, y).cpu().numpy()
    assert np.array_equal(got, again)            # deterministic (no atomics on the path)


def test_teacher_forced_steps_bf16_vs_reference_trajectory(golden_dir):
    """DDIM and DDPM+CFG teacher-forced steps in bf16 against the reference's fp32 trajectories: x_{t-1} within 3e-2."""
    from tests.cases import SAMPLER_CASES
    for name in ('ddim_eta0_li', 'ddpm_cfg', 'ddpm_learned'):
        case = SAMPLER_CASES[name]
        g = np.load(os.path.join(golden_dir, 'sampler_{}.npz'.format(name)))
        cfg = dict(TINY_CFGS[case['cfg']])
        cfg['out_channels'] = cfg['in_channels'] * 2
        m = build(cfg, seed=case.get('wseed', 99), sigma_zero=case.get('sigma_zero', 0.005))
        d = Diffusion(m, 1000, case['S'], case['var'], 'simple', beta_schedule=case['sched'],
                      guidance_method=case.get('guidance'), guidance_strength=case.get('w'), use_ddim=case['ddim'],
                      ddim_eta=case.get('eta'), device=torch.device(DEV))
        y = torch.from_numpy(g['y']).to(DEV) if 'y' in g.files else None
        noises = torch.from_numpy(g['noises'])
        traj = g['traj']
        S = case['S']
        x = torch.from_numpy(g['xT'])
        worst = 0.0
        for i, t in enumerate(reversed(range(S))):
            got = d.denoise(x=x, kwargs={'y': y} if y is not None else {}, batch_size=x.shape[0], steps_to_do=1, first_index=t,
                            progress=False, noise=noises).cpu().numpy()
            worst = max(worst, float(np.abs(got - traj[i]).max()))
            x = torch.from_numpy(traj[i])
        print('bf16 teacher-forced {}: max |x_(t-1) - reference| {:.2e}'.format(name, worst))
        assert worst < TEACHER_FORCED_BF16_BOUND, (name, worst)


# Measured drift of the FREE-RUNNING bf16 chains against the reference's fp32 trajectories (MI355X; printed by the tests below).
# Every bound in this file is <= 2x what was measured, so a regression of the bf16 path shows up here, not only a broken kernel.
# max |x_t - reference x_t| over the 10 steps; measured 0.037 / 0.037 / 0.028 / 0.023 / 0.0075 / 0.0066 / 0.015 / 0.011
FREE_RUNNING_BF16_BOUND = {'ddim_cfg': 0.07, 'ddim_eta05_learned': 0.07, 'ddim_eta0_li': 0.055, 'ddpm_li': 0.045,
                           'ddpm_learned': 0.015, 'ddpm_small': 0.013, 'ddpm_large': 0.03, 'ddpm_cfg': 0.021}
# forward, relative rms / max over absmax against the reference's fp32 output; measured (rms, max):
#   adagn_updown 1.55e-2, 1.67e-2 | plain_convres_legacy 1.51e-2, 2.81e-2 | pool_resample 1.38e-2, 1.57e-2 | odd_sizes 2.43e-2, 4.38e-2
TINY_FWD_BF16_BOUND = {'adagn_updown': (3.0e-2, 3.3e-2), 'plain_convres_legacy': (3.0e-2, 5.6e-2), 'pool_resample': (2.7e-2, 3.1e-2),
                       'odd_sizes': (4.8e-2, 8.0e-2)}
#   EMNIST 1.50e-2, 1.67e-2 | OPENAI_64 1.00e-2, 1.14e-2 | OPENAI_128 1.06e-2, 1.31e-2 | OPENAI_256 8.8e-3, 9.6e-3
PRESET_FWD_BF16_BOUND = {'EMNIST': (3.0e-2, 3.3e-2), 'OPENAI_64': (2.0e-2, 2.3e-2), 'OPENAI_128': (2.1e-2, 2.6e-2), 'OPENAI_256': (1.76e-2, 1.9e-2)}
TAP_BF16_BOUND = 4e-2                     # per-block intermediates, relative rms; worst measured 1.42e-2 / 1.26e-2 / 1.39e-2 / 2.08e-2
TEACHER_FORCED_BF16_BOUND = 1.2e-2        # measured 4.7e-3 / 6.1e-3 / 4.7e-3 (tiny models, 10 steps each)
CONFIG1_STEP_BF16_BOUND = 1.35e-2        # measured 6.8e-3 (EMNIST preset, every 5th of the 50 DDIM steps, against the fp32 HIP trajectory)
CONFIG4_STEP_BF16_BOUND = 3.0e-4          # measured 1.4e-4 / 1.3e-4 (128x128 DDPM + CFG step against the CPU oracle)


@pytest.mark.parametrize('name', sorted(__import__('tests.cases', fromlist=['SAMPLER_CASES']).SAMPLER_CASES))
def test_free_running_chain_bf16_vs_reference_trajectory(golden_dir, name):
    """The eight committed 10-step sampler trajectories of the REFERENCE (contractive synthetic weights, the reference's own
    noise draws; DDIM eta 0 / 0.5, DDPM x 4 variance types, classifier-free guidance) run FREE in bf16: every x_t feeds the
    next step, nothing is teacher-forced (diffusion.py:215-220,266-369).  Checked along the whole trajectory."""
    from tests.cases import SAMPLER_CASES
    case = SAMPLER_CASES[name]
    g = np.load(os.path.join(golden_dir, 'sampler_{}.npz'.format(name)))
    cfg = dict(TINY_CFGS[case['cfg']])
    learned = case['var'] in ('learned', 'learned_interpolation')
    cfg['out_channels'] = cfg['in_channels'] * (2 if learned else 1)
    m = build(cfg, seed=case.get('wseed', 99), sigma_zero=case.get('sigma_zero', 0.005))
    S = case['S']
    d = Diffusion(m, 1000, S, case['var'], 'simple', beta_schedule=case['sched'], guidance_method=case.get('guidance'),
                  guidance_strength=case.get('w'), use_ddim=case['ddim'], ddim_eta=case.get('eta'), device=torch.device(DEV))
    y = torch.from_numpy(g['y']).to(DEV) if 'y' in g.files else None
    kwargs = {'y': y} if y is not None else None
    noises, traj, xT = torch.from_numpy(g['noises']), g['traj'], torch.from_numpy(g['xT'])
    tr = []
    out = d.denoise(x=xT, kwargs=kwargs, batch_size=xT.shape[0], progress=False, noise=noises, trace=tr)
    got = torch.stack(tr).cpu().numpy()
    per_step = np.abs(got - traj).reshape(S, -1).max(1)
    rms, mx = _errs(got[-1], traj[-1])
    print('bf16 free-running {}: max |x_t - reference| per step {}  final rel rms {:.2e} max/absmax {:.2e}'.format(
        name, np.array2string(per_step, precision=4), rms, mx))
    bound = FREE_RUNNING_BF16_BOUND[name]
    assert per_step.max() < bound, (name, per_step.max(), bound)
    d.use_graph = True
    again = d.denoise(x=xT, kwargs=kwargs, batch_size=xT.shape[0], progress=False, noise=noises)
    assert torch.equal(again, out)


def test_config1_emnist_ddim50_bf16(golden_dir):
    """BASELINE configs[0] (EMNIST preset, 50-step DDIM, batch 4) in bf16.  (a) Every 5th step teacher-forced from the fp32
    HIP trajectory (which test_gpu_model pins to the reference's CPU output at 1e-3): x_{t-1} within 2x the measured
    6.8e-3 of the fp32 step.  (b) The chain run FREE against the REFERENCE's output: with these synthetic, non-contractive
    weights that per-step error grows to O(1) over 50 steps (measured: max 1.07, relative rms 0.34) -- stated, not hidden;
    the bound only says the chain stays on the data range's scale."""
    g = np.load(os.path.join(golden_dir, 'config1_emnist_ddim50.npz'))
    kw = dict(beta_schedule='cosine', use_ddim=True, ddim_eta=0.0)
    xT, y = torch.from_numpy(g['xT']), torch.from_numpy(g['y']).to(DEV)
    m32 = build(dict(DA.EMNIST_MODEL_ARGS), dtype='fp32')
    d32 = Diffusion(m32, 1000, 50, 'learned_interpolation', 'hybrid', device=torch.device(DEV), **kw)
    tr = []
    out32 = d32.denoise(x=xT, kwargs={'y': y}, batch_size=4, progress=False, trace=tr)
    assert np.abs(out32.cpu().numpy() - g['out']).max() < 1e-3
    m = build(dict(DA.EMNIST_MODEL_ARGS))
    d = Diffusion(m, 1000, 50, 'learned_interpolation', 'hybrid', device=torch.device(DEV), **kw)
    worst = 0.0
    for i, t in enumerate(reversed(range(50))):
        if i % 5 and i != 49:
            continue
        xt = xT if i == 0 else tr[i - 1].cpu()
        got = d.denoise(x=xt, kwargs={'y': y}, batch_size=4, steps_to_do=1, first_index=t, progress=False)
        worst = max(worst, float((got - tr[i]).abs().max()))
    out = d.denoise(x=xT, kwargs={'y': y}, batch_size=4, progress=False).cpu().numpy()
    err = float(np.abs(out - g['out']).max())
    rms, mx = _errs(out, g['out'])
    print('bf16 config[0]: teacher-forced step max err {:.2e}; 50-step free-running vs reference: max {:.2e}, rel rms {:.2e}'.format(
        worst, err, rms))
    assert worst < CONFIG1_STEP_BF16_BOUND, worst
    assert np.isfinite(out).all() and rms < 0.7, (err, rms)


def test_config4_workload_bf16_ddpm_cfg_128():
    """BASELINE configs[3] in bf16: 128x128 preset, num_classes=1001, DDPM, CFG 0.8 -- teacher-forced steps at B=1 against
    the fp32 ORACLE (CPU restatement of the reference) with injected noise, graph replay == eager, deterministic."""
    margs = dict(DA.OPENAI_128_MODEL_ARGS)
    margs['num_classes'] = 1001
    m = build(margs)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    kw = dict(beta_schedule='linear', use_ddim=False, guidance_method='classifier_free', guidance_strength=0.8)
    d = Diffusion(m, 1000, 1000, 'learned_interpolation', 'hybrid', device=torch.device(DEV), **kw)
    so = DO.SamplerOracle(lambda xx, tt, yy: UO.unet_forward(sd, margs, xx, tt, yy), DO.Schedule(1000, 1000, 'linear'),
                          'learned_interpolation', use_ddim=False, guidance_method='classifier_free', guidance_strength=0.8)
    torch.manual_seed(0)
    x = torch.randn(1, 3, 128, 128)
    y = torch.tensor([417])
    for t in (999, 0):
        nz = torch.randn(1, 3, 128, 128)
        noises = torch.zeros(t + 1, 1, 3, 128, 128)
        noises[t] = nz
        got = d.denoise(x=x, kwargs={'y': y.to(DEV)}, batch_size=1, steps_to_do=1, first_index=t, progress=False,
                        noise=noises).cpu()
        ref, _ = so.ddpm_step(x, t, y, nz)
        err = (got - ref).abs().max().item()
        print('bf16 config[3] teacher-forced step t={}: max err {:.2e}'.format(t, err))
        assert err < CONFIG4_STEP_BF16_BOUND, (t, err)
        del noises
    B = 8
    xb = torch.randn(B, 3, 128, 128)
    yb = (torch.arange(B) * 37) % 1000 + 1
    d.seed = 123
    a = d.denoise(x=xb, kwargs={'y': yb.to(DEV)}, batch_size=B, steps_to_do=3, progress=False)
    b = d.denoise(x=xb, kwargs={'y': yb.to(DEV)}, batch_size=B, steps_to_do=3, progress=False)
    d.use_graph = False
    c = d.denoise(x=xb, kwargs={'y': yb.to(DEV)}, batch_size=B, steps_to_do=3, progress=False)
    assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c)


def test_sample_cli_in_bf16(tmp_path, monkeypatch):
    """scripts/sample.py with ND_COMPUTE_DTYPE=bf16 (the reference's CLI has no precision flag for sampling): same files,
    images within two grey levels of the fp32 run."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'nice-diffusion_amd', 'scripts'))
    import sample
    cfg = dict(TINY_CFGS['adagn_updown'])
    sd = UO.synth_state_dict(cfg, seed=11)
    ckpt = str(tmp_path / 'tiny_model.pt')
    torch.save(sd, ckpt)
    outs = {}
    for mode in ('fp32', 'bf16'):
        out_dir = str(tmp_path / mode) + '/'
        os.makedirs(out_dir)
        monkeypatch.setenv('ND_COMPUTE_DTYPE', mode)
        argv = ['--model_path', ckpt, '--custom', '--batch_size', '2', '--num_samples', '1', '--resolution', '16',
                '--model_channels', '32', '--channel_mult', '1/2', '--num_res_blocks', '1', '--attention_resolutions', '8',
                '--num_classes', '10', '--num_head_channels', '32', '--split_qkv_first', '--resblock_updown', '--use_adaptive_gn',
                '--rescaled_num_steps', '5', '--beta_schedule', 'cosine', '--sampling_var_type', 'learned_interpolation',
                '--use_ddim', '--ddim_eta', '0.0', '--seed', '0', '--labels', '3', '--save_path', out_dir]
        sample.main(argv)
        assert sorted(os.listdir(out_dir)) == ['3_sample0.jpg', '3_sample1.jpg']
        # the same samples through the library, for a numeric comparison (JPEG bytes are lossy)
        m = build(cfg, seed=11, dtype=mode)
        d = Diffusion(m, 1000, 5, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=True, ddim_eta=0.0,
                      device=torch.device(DEV))
        torch.manual_seed(0)
        xT = torch.randn(2, 3, 16, 16)
        outs[mode] = sample.saved_bytes(d.denoise(x=xT, kwargs={'y': torch.tensor([3, 3]).to(DEV)}, batch_size=2,
                                                  progress=False), 3).astype(int)
    monkeypatch.delenv('ND_COMPUTE_DTYPE')
    assert np.abs(outs['fp32'] - outs['bf16']).max() <= 2


@pytest.mark.parametrize('pname,B,cfg', [('OPENAI_128', 16, True), ('OPENAI_256', 16, False)])
def test_full_size_properties_bf16(golden_dir, pname, B, cfg):
    """BASELINE configs[3] / [4] at their per-GPU batch in bf16 (the plans bench.py --workload config4 / config5 time).
    (0) two rows of the full-batch plan's output vs the REAL reference's fp32 output for those rows
    (tests/golden/config{4,5}_fullbatch_rows.npz) within the bf16 bound (<= 2x the measured value);
    (a) rows are independent: row r of the full-batch forward equals the same row run in a batch of 2 to within the bf16
    tolerance (different batch sizes pick different tile variants, i.e. summation orders);  (b) two sampler steps are
    finite and bitwise repeatable, graph replay included."""
    margs = dict(getattr(DA, pname + '_MODEL_ARGS'))
    if cfg:
        margs['num_classes'] += 1
    from tests.test_gpu_model import full_batch_forward, _preload_committed_tune_cache
    # the committed kernel choices of the bench workload (asserted to be taken in full): the plan below IS the timed one
    _preload_committed_tune_cache('config4' if cfg else 'config5')
    m = build(margs)
    R = margs['resolution']
    NI = 2 * B if cfg else B
    g = np.load(os.path.join(golden_dir, '{}_fullbatch_rows.npz'.format('config4' if cfg else 'config5')))
    gst = int(g['stride'])
    ref_rows = full_batch_forward(m, R, B, cfg, int(g['t'][0]))[torch.from_numpy(g['rows'])].cpu().numpy()
    rms0, mx0 = _errs(ref_rows[:, :, ::gst, ::gst], g['out_sub'])
    print(pname, 'bf16 full-batch rows vs the fp32 reference: rel rms {:.3e}, max/absmax {:.3e}'.format(rms0, mx0))
    assert rms0 < 1.9e-2 and mx0 < 2.3e-2, (rms0, mx0)         # measured 9.6e-3 / 1.14e-2 (128x128), 8.9e-3 / 9.5e-3 (256x256)
    torch.manual_seed(0)
    x = torch.randn(NI, 3, R, R)
    y = (torch.arange(NI) * 37) % 1000 + 1
    t = torch.full((NI,), 498)
    big = m(x.to(DEV), t.to(DEV), y.to(DEV))
    assert torch.isfinite(big).all()
    idx = torch.tensor([0, NI - 1])
    small = m(x[idx].to(DEV), t[idx].to(DEV), y[idx].to(DEV))
    rms, mx = _errs(big[idx].cpu().numpy(), small.cpu().numpy())
    assert rms < 2e-2 and mx < 5e-2, (rms, mx)
    kw = dict(beta_schedule='linear', use_ddim=False, guidance_method='classifier_free', guidance_strength=0.8) if cfg else \
        dict(beta_schedule='linear', use_ddim=True, ddim_eta=0.0)
    d = Diffusion(m, 1000, 50, 'learned_interpolation', 'hybrid', device=torch.device(DEV), **kw)
    d.seed = 7
    a = d.denoise(x=x[:B], kwargs={'y': y[:B].to(DEV)}, batch_size=B, steps_to_do=2, progress=False)
    b = d.denoise(x=x[:B], kwargs={'y': y[:B].to(DEV)}, batch_size=B, steps_to_do=2, progress=False)
    assert torch.isfinite(a).all() and torch.equal(a, b)
