#!/usr/bin/env python3
"""Interleaved A/B of Winograd variants in ONE process (clock and thermal state shared):
   python tools/ab_variants.py "NI H W C N;..." 8,12 [rounds]   -> median ms / algorithmic TFLOP/s per variant"""
import sys, os, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
shapes = [tuple(int(v) for v in s.split()) for s in sys.argv[1].split(';') if s.strip()]
variants = [int(v) for v in sys.argv[2].split(',')]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
warm = False
for (NI, H, W, C, N) in shapes:
    torch.manual_seed(0)
    x = torch.randn(NI * H * W * C, device='cuda'); w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
    b = torch.randn(N, device='cuda'); out = torch.empty(NI * H * W * N, device='cuda')
    w = torch.empty(lib.nd_conv_winograd_weight_floats(N, C), device='cuda')
    assert lib.nd_repack_conv_weight_winograd(w0.data_ptr(), w.data_ptr(), N, C, st) == 0
    fl = 2.0 * NI * H * W * N * 9 * C
    def run(v, n):
        for _ in range(n):
            assert lib.nd_conv3x3_winograd_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0,
                                                out.data_ptr(), N, NI, H, W, N, 0, v, None, None, 0, st) == 0, _hip.last_error()
    if not warm:
        t0 = time.time()
        while time.time() - t0 < 2.0:
            run(variants[0], 10); torch.cuda.synchronize()
        warm = True
    res = {v: [] for v in variants}
    outs = {}
    for r in range(rounds):
        for v in variants:
            run(v, 2); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(v, 10); e1.record(); e1.synchronize()
            res[v].append(e0.elapsed_time(e1) / 10)
            if r == 0: outs[v] = out.clone()
    same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants)
    print((NI, H, W, C, N), '  '.join('v%d: %.3f ms %.1f TF' % (v, statistics.median(res[v]), fl / statistics.median(res[v]) / 1e9) for v in variants),
          ' bit-identical' if same else ' DIFFERENT BITS')
