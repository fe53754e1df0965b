#!/usr/bin/env python3
"""Interleaved A/B of library BUILDS on the bf16 3x3 convolution (nd_conv3x3_bf16_stats_nhwc / nd_conv_bf16_nhwc), one process:
    python tools/ab_bf16.py libA.so,libB.so "NI H W C N;..." [rounds] [stats|plain] [variant]
The builds' outputs (and statistics rows) are compared bit for bit."""
import ctypes, sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
libs = sys.argv[1].split(',')
shapes = [tuple(int(v) for v in s.split()) for s in sys.argv[2].split(';') if s.strip()]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
mode = sys.argv[4] if len(sys.argv) > 4 else 'stats'
var = int(sys.argv[5]) if len(sys.argv) > 5 else 11
L = []
for path in libs:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, at in _hip.SIGNATURES.items():
        if hasattr(l, name):
            getattr(l, name).argtypes = at; getattr(l, name).restype = ctypes.c_int
    l.nd_conv_bf16_weight_elems.argtypes = [ctypes.c_int] * 3; l.nd_conv_bf16_weight_elems.restype = ctypes.c_int64
    L.append(l)
st = torch.cuda.current_stream().cuda_stream
for (NI, H, W, C, N) in shapes:
    torch.manual_seed(0)
    x = torch.randn(NI * H * W * C, device='cuda').to(torch.bfloat16)
    w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
    b = torch.randn(N, device='cuda')
    res = torch.randn(NI * H * W * N, device='cuda').to(torch.bfloat16)
    ws, outs, css = [], [], []
    for l in L:
        w = torch.empty(l.nd_conv_bf16_weight_elems(N, C, 3), dtype=torch.bfloat16, device='cuda')
        assert l.nd_repack_conv_weight_bf16(w0.data_ptr(), w.data_ptr(), N, C, 3, 0, st) == 0
        ws.append(w)
        outs.append(torch.empty(NI * H * W * N, dtype=torch.bfloat16, device='cuda'))
        rows = l.nd_conv_bf16_stats_rows(NI, H, W, N, var) if mode == 'stats' else 0
        css.append(torch.zeros(NI * max(rows, 1) * 2 * N, device='cuda') if rows > 0 else None)
    fl = 2.0 * NI * H * W * N * 9 * C
    def run(i, n):
        for _ in range(n):
            if css[i] is not None:
                rc = L[i].nd_conv3x3_bf16_stats_nhwc(x.data_ptr(), C, C, None, 0, 0, ws[i].data_ptr(), b.data_ptr(), None, N, res.data_ptr(), N,
                                                     outs[i].data_ptr(), N, NI, H, W, N, 0, var, None, None, C, css[i].data_ptr(), st)
            else:
                rc = L[i].nd_conv_bf16_nhwc(x.data_ptr(), C, C, None, 0, 0, ws[i].data_ptr(), b.data_ptr(), None, N, res.data_ptr(), N,
                                            outs[i].data_ptr(), N, NI, H, W, N, 3, 0, var, None, None, C, st)
            assert rc == 0, rc
    for i in range(len(L)):
        run(i, 20)
    torch.cuda.synchronize()
    t = [[] for _ in L]
    for r in range(rounds):
        for i in range(len(L)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(i, 10); e1.record(); e1.synchronize()
            t[i].append(e0.elapsed_time(e1) / 10)
    same = all(torch.equal(outs[0], o) for o in outs[1:]) and all((css[0] is None) or torch.equal(css[0], c) for c in css[1:])
    print((NI, H, W, C, N), mode, '  '.join('%s: %.4f ms %.0f TF' % (os.path.basename(libs[i])[6:-3] or 'base', statistics.median(t[i]),
                                                                       fl / statistics.median(t[i]) / 1e9) for i in range(len(L))),
          '  bit-equal' if same else '  OUTPUTS DIFFER')
