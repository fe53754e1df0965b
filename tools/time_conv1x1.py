#!/usr/bin/env python3
"""Timing of the fp32 1x1 forms on one shape (GPU box): nd_conv_nhwc variant v with / without a residual, and
nd_conv1x1_stats_nhwc with / without one.   python tools/time_conv1x1.py "NI H W C N;..." [variants, default 14,15]"""
import sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
lib = _hip.load()
shapes = [tuple(int(v) for v in s.split()) for s in sys.argv[1].split(';') if s.strip()]
variants = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [14, 15]
st = torch.cuda.current_stream().cuda_stream
a = torch.randn(4096, 4096, device='cuda')
for _ in range(40):
    a @ a
torch.cuda.synchronize()


def timed(fn, n=10, rounds=5):
    fn(); torch.cuda.synchronize()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return statistics.median(res)


for (NI, H, W, C, N) in shapes:
    torch.manual_seed(0)
    M = NI * H * W
    x = torch.randn(M * C, device='cuda'); w0 = torch.randn(N, C, device='cuda') * 0.05
    b = torch.randn(N, device='cuda'); out = torch.empty(M * N, device='cuda'); res = torch.randn(M * N, device='cuda')
    wp = torch.empty(lib.nd_conv_weight_floats(N, C, 1), device='cuda')
    assert lib.nd_repack_conv_weight(w0.data_ptr(), wp.data_ptr(), N, C, 1, st) == 0
    rows = lib.nd_conv1x1_stats_rows(NI, H, W, N)
    cs = torch.empty(max(1, NI * rows * 2 * N), device='cuda')
    fl = 2.0 * M * N * C
    line = []
    for v in variants:
        for r in (None, res):
            def f():
                assert lib.nd_conv_nhwc(x.data_ptr(), C, C, None, 0, 0, wp.data_ptr(), b.data_ptr(), None, 0,
                                        None if r is None else r.data_ptr(), 0 if r is None else N, out.data_ptr(), N, 1, 1, M, N, 1, 0, v,
                                        None, None, 0, st) == 0
            ms = timed(f)
            line.append('v%d%s %.1f us %.0f TF' % (v, '' if r is None else '+res', ms * 1e3, fl / ms / 1e9))
    if rows > 0:
        for r in (None, res):
            def f():
                assert lib.nd_conv1x1_stats_nhwc(x.data_ptr(), C, C, None, 0, 0, wp.data_ptr(), b.data_ptr(), None if r is None else r.data_ptr(),
                                                 0 if r is None else N, out.data_ptr(), N, NI, H, W, N, 0, cs.data_ptr(), st) == 0
            ms = timed(f)
            line.append('stats%s %.1f us %.0f TF' % ('' if r is None else '+res', ms * 1e3, fl / ms / 1e9))
    print((NI, H, W, C, N), '  '.join(line), flush=True)
