#!/usr/bin/env python3
"""conv_wf4_kernel (Winograd F(4x4,3x3), nd_conv3x3_winograd_f4_nhwc) on a GPU box: parity against F.conv2d on a sweep of
shapes and fused options, run-to-run bit identity, and interleaved timing against conv_wino4_kernel (F(2x2,3x3)).

    python tools/wf4_check.py            # parity sweep + timing of the headline shapes
    python tools/wf4_check.py time       # timing only"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
import torch.nn.functional as F
from nicediffusion import _hip

DEV = 'cuda'
L = _hip.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def pack_f4(w):
    N, C = w.shape[0], w.shape[1]
    n = L.nd_conv_winograd_f4_weight_floats(0, N, C)
    out = torch.full((n,), float('nan'), device=DEV)
    _hip.check(L.nd_repack_conv_weight_winograd_f4(w.contiguous().to(DEV).data_ptr(), out.data_ptr(), N, C, 0, st()))
    return out


def pack_f2(w):
    N, C = w.shape[0], w.shape[1]
    n = L.nd_conv_winograd_weight_floats(N, C)
    out = torch.empty(n, device=DEV)
    _hip.check(L.nd_repack_conv_weight_winograd(w.contiguous().to(DEV).data_ptr(), out.data_ptr(), N, C, st()))
    return out


def run_f4(xd, Cin, ld, wd, bd, rbd, resd, ldr, B, H, W, N, flags=0, stats=None, splits=1, ws=None, ldo=None, out=None):
    ldo = ldo or N
    if out is None:
        out = torch.full((B * H * W * ldo,), float('nan'), device=DEV)
    rc = L.nd_conv3x3_winograd_f4_nhwc(xd.data_ptr(), Cin, ld, wd.data_ptr(), None if bd is None else bd.data_ptr(),
                                       None if rbd is None else rbd.data_ptr(), 0 if rbd is None else N,
                                       None if resd is None else resd.data_ptr(), ldr, out.data_ptr(), ldo, B, H, W, N, flags, 0,
                                       None if stats is None else stats.data_ptr(), splits, None if ws is None else ws.data_ptr(), st())
    return rc, out


def parity():
    worst = 0.0
    cases = [(1, 32, 48, 16, 16), (2, 64, 48, 16, 16), (2, 64, 96, 32, 32), (1, 96, 192, 64, 64), (4, 32, 48, 8, 8), (6, 64, 100, 8, 8),
             (2, 32, 40, 28, 28), (3, 128, 52, 12, 20), (1, 192, 192, 32, 32), (8, 160, 96, 8, 8), (2, 64, 6, 16, 16)]
    for (B, Cin, N, H, W) in cases:
        x, w, b = rnd(B, Cin, H, W, seed=1), rnd(N, Cin, 3, 3, seed=2, scale=0.05), rnd(N, seed=3)
        rb, res = rnd(B, N, seed=5), rnd(B, N, H, W, seed=6)
        ref0 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        xd, wd, bd, rbd, resd = nhwc(x), pack_f4(w), b.to(DEV), rb.to(DEV), nhwc(res)
        # plain
        rc, out = run_f4(xd, Cin, Cin, wd, bd, None, None, 0, B, H, W, N)
        assert rc == 0, _hip.last_error()
        got = out.view(B, H, W, N).permute(0, 3, 1, 2).cpu().double()
        e0 = (got - ref0).abs().max().item()
        # bias + per-image bias + residual + statistics
        rows = L.nd_conv_winograd_f4_stats_rows(0, B, H, W)
        stats = torch.full((B * rows * 2 * N,), float('nan'), device=DEV)
        rc, out = run_f4(xd, Cin, Cin, wd, bd, rbd, resd, N, B, H, W, N, stats=stats)
        assert rc == 0, _hip.last_error()
        ref1 = ref0 + rb.double()[:, :, None, None] + res.double()
        got1 = out.view(B, H, W, N).permute(0, 3, 1, 2).cpu().double()
        e1 = (got1 - ref1).abs().max().item()
        sg = stats.view(B, rows, 2, N).cpu().double().sum(1)
        es = (sg[:, 0] - got1.sum((2, 3))).abs().max().item() / (H * W) ** 0.5
        eq = (sg[:, 1] - (got1 * got1).sum((2, 3))).abs().max().item() / (H * W)
        # run to run
        rc, out2 = run_f4(xd, Cin, Cin, wd, bd, rbd, resd, N, B, H, W, N, stats=stats)
        same = torch.equal(out, out2)
        # SiLU out, 2x-upsampled input
        e2 = e3 = float('nan')
        rc, o3 = run_f4(xd, Cin, Cin, wd, bd, None, None, 0, B, H, W, N, flags=_hip.CONV_SILU_OUT)
        assert rc == 0, _hip.last_error()
        e2 = (o3.view(B, H, W, N).permute(0, 3, 1, 2).cpu().double() - F.silu(ref0)).abs().max().item()
        if H % 8 == 0 and W % 8 == 0 and (H >= 24 or H == 16):
            xs = rnd(B, Cin, H // 2, W // 2, seed=7)
            refu = F.conv2d(F.interpolate(xs, scale_factor=2, mode='nearest').double(), w.double(), b.double(), padding=1)
            rc, o4 = run_f4(nhwc(xs), Cin, Cin, wd, bd, None, None, 0, B, H, W, N, flags=_hip.CONV_IN_UP2X)
            assert rc == 0, _hip.last_error()
            e3 = (o4.view(B, H, W, N).permute(0, 3, 1, 2).cpu().double() - refu).abs().max().item()
        # split over K
        e4 = float('nan')
        if Cin >= 64 and N % 4 == 0:
            S = 2
            need = L.nd_conv_splitk_workspace_floats(B, H, W, N, Cin, 3, S)
            ws = torch.full((max(need, 4),), float('nan'), device=DEV)
            rc, o5 = run_f4(xd, Cin, Cin, wd, bd, rbd, resd, N, B, H, W, N, splits=S, ws=ws)
            assert rc == 0, _hip.last_error()
            e4 = (o5.view(B, H, W, N).permute(0, 3, 1, 2).cpu().double() - ref1).abs().max().item()
        print('B%d Cin%d N%d %dx%d: plain %.2e  fused %.2e  stats %.1e/%.1e  silu %.2e  up2x %.2e  splitK %.2e  repeat %s' %
              (B, Cin, N, H, W, e0, e1, es, eq, e2, e3, e4, same), flush=True)
        worst = max(worst, e0, e1, e2, 0 if e3 != e3 else e3, 0 if e4 != e4 else e4)
        assert same and torch.isfinite(out).all()
    print('worst abs error %.3e (|y| ~ %.1f)' % (worst, ref0.abs().max().item()))
    return worst


def timeit(fn, n=6, bursts=3):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(bursts):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / n
        best = t if best is None else min(best, t)
    return best


def timing():
    names = [L.nd_conv_winograd_variant_name(v) for v in range(L.nd_conv_winograd_num_variants())]
    v4 = names.index(b'nd::conv_wino4_kernel')
    shapes = [(64, 192, 192, 64, 64), (64, 384, 192, 64, 64), (64, 384, 384, 32, 32), (64, 768, 384, 32, 32), (64, 576, 576, 16, 16),
              (64, 1152, 576, 16, 16), (64, 768, 768, 8, 8), (64, 1536, 768, 8, 8)]
    # warm the clock
    a = torch.randn(4096, 4096, device=DEV)
    for _ in range(30):
        a @ a
    torch.cuda.synchronize()
    for (B, Cin, N, H, W) in shapes:
        x = torch.randn(B * H * W * Cin, device=DEV)
        w = rnd(N, Cin, 3, 3, seed=2, scale=0.05)
        w4, w2 = pack_f4(w), pack_f2(w)
        b = torch.randn(N, device=DEV)
        out = torch.empty(B * H * W * N, device=DEV)
        rows4 = L.nd_conv_winograd_f4_stats_rows(0, B, H, W)
        rows2 = L.nd_conv_winograd_stats_rows(v4, B, H, W)
        stats = torch.empty(B * max(rows4, rows2) * 2 * N, device=DEV)

        def f4(stats_=None, splits=1, ws=None):
            rc, _ = run_f4(x, Cin, Cin, w4, b, None, None, 0, B, H, W, N, stats=stats_, splits=splits, ws=ws, out=out)
            assert rc == 0, _hip.last_error()

        def f2():
            rc = L.nd_conv3x3_winograd_vstats_nhwc(x.data_ptr(), Cin, Cin, None, 0, 0, w2.data_ptr(), b.data_ptr(), None, 0, None, 0,
                                                   out.data_ptr(), N, B, H, W, N, 0, v4, stats.data_ptr(), st())
            assert rc == 0, _hip.last_error()
        res = []
        for _ in range(2):
            res.append((timeit(f2), timeit(lambda: f4(stats)), timeit(f4)))
        t2 = min(r[0] for r in res); t4s = min(r[1] for r in res); t4 = min(r[2] for r in res)
        fl = 2.0 * B * H * W * N * 9 * Cin
        line = 'B%d %dx%d %d->%d: F2+stats %.1f us (%.0f TF/s exec)  F4+stats %.1f us  F4 %.1f us (%.0f TF/s exec, %.2f of peak)  ratio %.3f' % (
            B, H, W, Cin, N, t2 * 1e3, fl * 4 / 9 / t2 / 1e9, t4s * 1e3, t4 * 1e3, fl / 4 / t4 / 1e9, fl / 4 / t4 / 1e9 / 157.3, t4s / t2)
        if B * H * W <= 16384 * 4:
            for S in (2, 4):
                need = L.nd_conv_splitk_workspace_floats(B, H, W, N, Cin, 3, S)
                ws = torch.empty(max(need, 4), device=DEV)
                line += '  splitK%d %.1f us' % (S, timeit(lambda: f4(None, S, ws)) * 1e3)
        print(line, flush=True)


if __name__ == '__main__':
    if len(sys.argv) < 2 or sys.argv[1] != 'time':
        parity()
    timing()
