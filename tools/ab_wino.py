#!/usr/bin/env python3
"""A/B of Winograd builds: python tools/ab_wino.py libA.so,libB.so "NI H W C N variant;..." [rounds]"""
import ctypes, sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
libs = sys.argv[1].split(',')
shapes = [tuple(int(v) for v in s.split()) for s in sys.argv[2].split(';') if s.strip()]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = []
for path in libs:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, at in _hip.SIGNATURES.items():
        getattr(l, name).argtypes = at; getattr(l, name).restype = ctypes.c_int
    l.nd_conv_winograd_weight_floats.argtypes = [ctypes.c_int] * 2; l.nd_conv_winograd_weight_floats.restype = ctypes.c_int64
    L.append(l)
st = torch.cuda.current_stream().cuda_stream
for (NI, H, W, C, N, var) in shapes:
    torch.manual_seed(0)
    x = torch.randn(NI * H * W * C, device='cuda'); w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
    b = torch.randn(N, device='cuda'); out = torch.empty(NI * H * W * N, device='cuda')
    ws = []
    for l in L:
        w = torch.empty(l.nd_conv_winograd_weight_floats(N, C), device='cuda')
        assert l.nd_repack_conv_weight_winograd(w0.data_ptr(), w.data_ptr(), N, C, st) == 0
        ws.append(w)
    fl = 2.0 * NI * H * W * N * 9 * C
    res = [[] for _ in L]
    def run(i, n):
        for _ in range(n):
            assert L[i].nd_conv3x3_winograd_nhwc(x.data_ptr(), C, C, None, 0, 0, ws[i].data_ptr(), b.data_ptr(), None, 0, None, 0, out.data_ptr(), N, NI, H, W, N, 0, var, None, None, 0, st) == 0
    for i in range(len(L)): run(i, 2)
    torch.cuda.synchronize()
    for r in range(rounds):
        for i in range(len(L)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(i, 5); e1.record(); e1.synchronize()
            res[i].append(e0.elapsed_time(e1) / 5)
    print((NI, H, W, C, N, var), '  '.join('%s: %.3f ms %.1f TF' % (os.path.basename(libs[i])[6:-3], statistics.median(res[i]), fl / statistics.median(res[i]) / 1e9) for i in range(len(L))))
