#!/usr/bin/env python3
"""What the vendor GEMM (hipBLASLt behind torch.matmul) sustains on this box, as a yardstick for the hand-written conv
kernels (it is NOT part of the product path): bf16 and fp32 GEMMs at a square size and at the implicit-GEMM shapes of the
dominant layers, each timed for ~1 s after a 1 s warm-up so that the clock has settled.
    python tools/gemm_ceiling.py"""
import time
import torch

dev = torch.device('cuda:0')
SHAPES = [('square 8192^3', 8192, 8192, 8192),
          ('128x128x32 img, 256->256 3x3 (M=524288,K=2304,N=256)', 524288, 256, 2304),
          ('64x64x32 img, 512->512 3x3 (M=131072,K=4608,N=512)', 131072, 512, 4608),
          ('64x64x64 img, 192->192 3x3 (M=262144,K=1728,N=192)', 262144, 192, 1728),
          ('32x32x64 img, 384->384 3x3 (M=65536,K=3456,N=384)', 65536, 384, 3456),
          # the 1x1 convolutions of configs[1] (short K)
          ('32x32x64 img, qkv 384->1152 1x1 (M=65536,K=384,N=1152)', 65536, 1152, 384),
          ('32x32x64 img, proj 384->384 1x1', 65536, 384, 384),
          ('64x64x64 img, skip 576->192 1x1', 262144, 192, 576),
          ('16x16x64 img, qkv 576->1728 1x1', 16384, 1728, 576),
          ('8x8x64 img, qkv 768->2304 1x1', 4096, 2304, 768)]
import sys
DTYPES = ((torch.bfloat16, 'bf16'), (torch.float32, 'fp32'))
if len(sys.argv) > 1:
    DTYPES = tuple(d for d in DTYPES if d[1] in sys.argv[1:])
for dt, name in DTYPES:
    for label, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(K, N, device=dev, dtype=dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        t0 = time.time()
        while time.time() - t0 < 1.0:
            torch.matmul(a, b, out=c)
            torch.cuda.synchronize()
        torch.matmul(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        e0.record()
        t0 = time.time()
        while time.time() - t0 < 1.0:
            for _ in range(4):
                torch.matmul(a, b, out=c)
            n += 4
            torch.cuda.current_stream().synchronize()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print('%s  %-58s %8.3f ms  %7.1f TFLOP/s' % (name, label, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        del a, b, c
