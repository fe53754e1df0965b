#!/usr/bin/env python3
"""Micro-benchmark of nd_conv_nhwc on one shape (GPU box only):
   python tools/conv_bench.py NI H W Cin N ksize [variant] [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
a = sys.argv[1:]
NI, H, W, C, N, ks = [int(v) for v in a[:6]]
variants = [int(v) for v in a[6].split(',')] if len(a) > 6 else [-1]
iters = int(a[7]) if len(a) > 7 else 20
lib = _hip.load()
dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(NI * H * W * C, device=dev)
w0 = torch.randn(N, C, ks, ks, device=dev) * 0.02
w = torch.empty(lib.nd_conv_weight_floats(N, C, ks), device=dev)
assert lib.nd_repack_conv_weight(w0.data_ptr(), w.data_ptr(), N, C, ks, torch.cuda.current_stream().cuda_stream) == 0
b = torch.randn(N, device=dev)
out = torch.empty(NI * H * W * N, device=dev)
st = torch.cuda.current_stream().cuda_stream
fl = 2.0 * NI * H * W * N * ks * ks * C
wino = os.environ.get('WINO') == '1'
res = torch.randn(NI * H * W * N, device=dev) if os.environ.get('RES') == '1' else None
resp = None if res is None else res.data_ptr()
if wino:
    w = torch.empty(lib.nd_conv_winograd_weight_floats(N, C), device=dev)
    assert lib.nd_repack_conv_weight_winograd(w0.data_ptr(), w.data_ptr(), N, C, st) == 0
    ref_out = torch.empty_like(out)
    wd = torch.empty(lib.nd_conv_weight_floats(N, C, 3), device=dev)
    assert lib.nd_repack_conv_weight(w0.data_ptr(), wd.data_ptr(), N, C, 3, st) == 0
    assert lib.nd_conv_nhwc(x.data_ptr(), C, C, None, 0, 0, wd.data_ptr(), b.data_ptr(), None, 0, None, 0, ref_out.data_ptr(), N, NI, H, W, N, 3, 0, -1, None, None, 0, st) == 0
import time
_warm = [False]
for v in variants:
    def run():
        if wino:
            rc = lib.nd_conv3x3_winograd_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), None, 0, resp, N, out.data_ptr(), N, NI, H, W, N, 0, v, None, None, 0, st)
            assert rc == 0, _hip.last_error()
            return
        rc = lib.nd_conv_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, out.data_ptr(), N,
                              NI, H, W, N, ks, 0, v, None, None, 0, st)
        assert rc == 0, _hip.last_error()
    try:
        run()
    except AssertionError as e:
        print('variant', v, 'n/a', e); continue
    torch.cuda.synchronize()
    if not _warm[0]:
        # the shader clock takes about a second of load to settle; without this the first variant measures 10 % slow
        t0 = time.time()
        while time.time() - t0 < 1.5:
            for _ in range(10):
                run()
            torch.cuda.synchronize()
        _warm[0] = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    if wino:
        print('   winograd vs direct max abs diff %.3e (ref absmax %.3f)' % ((out - ref_out - (0 if res is None else res)).abs().max().item(), ref_out.abs().max().item()))
    print('shape', (NI, H, W, C, N, ks), 'variant', v, 'auto->%d' % lib.nd_conv_select_variant(NI, H, W, N, ks, 0, 0) if v < 0 else '',
          '%.3f ms  %.1f TFLOP/s' % (ms, fl / ms / 1e9))
