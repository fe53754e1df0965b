#!/usr/bin/env python3
"""GroupNorm stats/apply micro-benchmark for PMC traffic calibration: python tools/gn_bench.py NI H W C iters"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
NI, H, W, C = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
x = torch.randn(NI * H * W * C, device='cuda'); out = torch.empty_like(x)
g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
stats = torch.zeros(NI * 64, dtype=torch.float64, device='cuda')
def run():
    stats.zero_()
    assert lib.nd_groupnorm_stats_nhwc(x.data_ptr(), C, C, None, 0, 0, None, 0, stats.data_ptr(), NI, H * W, 32, st) == 0
    assert lib.nd_groupnorm_apply_nhwc(x.data_ptr(), C, C, None, 0, 0, None, 0, stats.data_ptr(), g.data_ptr(), b.data_ptr(), None, None, 0, out.data_ptr(), C, NI, H, W, 32, 1e-5, 1, st) == 0
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
nbytes = x.numel() * 4
print('GN stats+apply %.3f ms; bytes read+read+write = %.1f MB -> %.2f TB/s' % (ms, 3 * nbytes / 1e6, 3 * nbytes / ms / 1e9))
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
ms = t(lambda: lib.nd_groupnorm_stats_nhwc(x.data_ptr(), C, C, None, 0, 0, None, 0, stats.data_ptr(), NI, H * W, 32, st))
print('   stats only %.4f ms -> %.2f TB/s (%.0f %% of 8 TB/s)' % (ms, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8 * 100))
ms = t(lambda: lib.nd_groupnorm_apply_nhwc(x.data_ptr(), C, C, None, 0, 0, None, 0, stats.data_ptr(), g.data_ptr(), b.data_ptr(), None, None, 0, out.data_ptr(), C, NI, H, W, 32, 1e-5, 1, st))
print('   apply only %.4f ms -> %.2f TB/s (%.0f %% of 8 TB/s)' % (ms, 2 * nbytes / ms / 1e9, 2 * nbytes / ms / 1e9 / 8 * 100))
