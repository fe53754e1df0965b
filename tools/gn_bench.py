#!/usr/bin/env python3
"""GroupNorm stats/apply micro-benchmark (also used for PMC traffic calibration):
    python tools/gn_bench.py NI H W C [iters] [bf16]
Reports each kernel's HBM rate against the 8 TB/s peak and checks that repeated statistics launches give the same bits."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
NI, H, W, C = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
bf16 = len(sys.argv) > 6 and sys.argv[6] == 'bf16'
dt = _hip.DT_BF16 if bf16 else _hip.DT_F32
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
x = torch.randn(NI * H * W * C, device='cuda').to(torch.bfloat16 if bf16 else torch.float32); out = torch.empty_like(x)
g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
nb = lib.nd_groupnorm_stats_blocks(NI, H * W, C, dt)
stats = torch.zeros(NI * nb * 64, dtype=torch.float64, device='cuda')
def f_stats():
    assert lib.nd_groupnorm_stats_nhwc(x.data_ptr(), C, C, None, 0, 0, None, 0, stats.data_ptr(), NI, H * W, 32, dt, st) == 0
def f_apply():
    assert lib.nd_groupnorm_apply_nhwc(x.data_ptr(), C, C, None, 0, 0, None, 0, stats.data_ptr(), nb, g.data_ptr(), b.data_ptr(), None, None, 0, out.data_ptr(), C, NI, H, W, 32, 1e-5, 1, dt, st) == 0
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
t0 = __import__('time').time()
while __import__('time').time() - t0 < 1.0:      # let the clock settle
    f_stats(); f_apply(); torch.cuda.synchronize()
f_stats(); torch.cuda.synchronize(); ref = stats.clone()
for _ in range(5):
    f_stats(); torch.cuda.synchronize()
    assert torch.equal(ref, stats), 'statistics are not reproducible'
nbytes = x.numel() * x.element_size()
ms = t(f_stats)
print('%s NI=%d %dx%d C=%d: stats %.4f ms -> %.2f TB/s (%.0f %% of 8 TB/s)' % ('bf16' if bf16 else 'fp32', NI, H, W, C, ms, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8 * 100))
ms = t(f_apply)
print('   apply %.4f ms -> %.2f TB/s (%.0f %% of 8 TB/s)' % (ms, 2 * nbytes / ms / 1e9, 2 * nbytes / ms / 1e9 / 8 * 100))
