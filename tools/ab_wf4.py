#!/usr/bin/env python3
"""Interleaved A/B of library BUILDS on conv_wf4_kernel (Winograd F(4x4,3x3)):
    python tools/ab_wf4.py libA.so,libB.so "NI H W C N;..." [rounds] [stats,res]
(libs from tools/build_one_variant.sh NAME nd_conv_winograd_f4.hip -D...; timing-only ablations give wrong results)"""
import ctypes, sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
libs = sys.argv[1].split(',')
shapes = [tuple(int(v) for v in s.split()) for s in sys.argv[2].split(';') if s.strip()]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
opts = sys.argv[4].split(',') if len(sys.argv) > 4 else []
want_stats = 'stats' in opts
want_res = 'res' in opts
SPLITS = int(os.environ.get('ND_AB_SPLITS', '1'))          # split over K (workspace + reduce pass, as the plan runs the small maps)          # residual + per-image bias: the out_conv of a residual block
L = []
for path in libs:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, at in _hip.SIGNATURES.items():
        if hasattr(l, name):
            getattr(l, name).argtypes = at; getattr(l, name).restype = ctypes.c_int
    l.nd_conv_winograd_f4_weight_floats.argtypes = [ctypes.c_int] * 3; l.nd_conv_winograd_f4_weight_floats.restype = ctypes.c_int64
    l.nd_conv_splitk_workspace_floats.argtypes = [ctypes.c_int] * 7; l.nd_conv_splitk_workspace_floats.restype = ctypes.c_int64
    L.append(l)
st = torch.cuda.current_stream().cuda_stream
a = torch.randn(4096, 4096, device='cuda')
for _ in range(40):
    a @ a
torch.cuda.synchronize()
for (NI, H, W, C, N) in shapes:
    torch.manual_seed(0)
    x = torch.randn(NI * H * W * C, device='cuda'); w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
    b = torch.randn(N, device='cuda'); out = torch.empty(NI * H * W * N, device='cuda')
    resid = torch.randn(NI * H * W * N, device='cuda') if want_res else None
    outs = []
    ws = []
    for l in L:
        w = torch.empty(l.nd_conv_winograd_f4_weight_floats(0, N, C), device='cuda')
        assert l.nd_repack_conv_weight_winograd_f4(w0.data_ptr(), w.data_ptr(), N, C, 0, st) == 0
        ws.append(w)
    rows = L[0].nd_conv_winograd_f4_stats_rows(0, NI, H, W)
    stats = torch.empty(NI * rows * 2 * N, device='cuda') if want_stats else None
    wsp = torch.empty(max(4, L[0].nd_conv_splitk_workspace_floats(NI, H, W, N, C, 3, SPLITS)), device='cuda') if SPLITS > 1 else None
    if SPLITS > 1:
        rows = L[0].nd_conv_winograd_f4_splitk_stats_rows(0, NI, H, W)
        stats = torch.empty(max(4, NI * rows * 2 * N), device='cuda') if want_stats and rows > 0 else None
    fl = 2.0 * NI * H * W * N * 9 * C / 4
    res = [[] for _ in L]
    def run(i, n):
        for _ in range(n):
            assert L[i].nd_conv3x3_winograd_f4_nhwc(x.data_ptr(), C, C, ws[i].data_ptr(), b.data_ptr(), None, 0, None if resid is None else resid.data_ptr(), N, out.data_ptr(), N,
                                                    NI, H, W, N, 0, 0, None if stats is None else stats.data_ptr(), SPLITS, None if wsp is None else wsp.data_ptr(), st) == 0
    for i in range(len(L)):
        run(i, 2)
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert os.environ.get('ND_AB_NOCHECK') or all(torch.equal(o, outs[0]) for o in outs), 'builds disagree (ND_AB_NOCHECK=1 for timing-only ablations)'
    for r in range(rounds):
        for i in range(len(L)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(i, 5); e1.record(); e1.synchronize()
            res[i].append(e0.elapsed_time(e1) / 5)
    print((NI, H, W, C, N), '  '.join('%s: %.1f us %.2f' % (os.path.basename(libs[i])[6:-3], statistics.median(res[i]) * 1e3,
                                                        fl / statistics.median(res[i]) / 1e9 / 157.3) for i in range(len(L))), flush=True)
