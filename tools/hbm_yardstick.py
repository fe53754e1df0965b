#!/usr/bin/env python3
"""HBM yardstick on this box: torch's device-to-device copy and an elementwise fp32 op on 200 MB / 600 MB tensors
(read + write bytes per second), next to which gn_apply_kernel's 5.3-5.5 TB/s is to be read."""
import torch
st = lambda: torch.cuda.synchronize()
for mb in (100, 200, 600, 1200):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    for name, fn in (('copy_', lambda: y.copy_(x)), ('mul', lambda: torch.mul(x, 1.5, out=y)), ('silu', lambda: torch.nn.functional.silu(x))):
        fn(); st()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print('%5d MB %-6s %.1f us  %.2f TB/s (read + write)' % (mb, name, best * 1e3, 2 * n * 4 / best / 1e9), flush=True)
