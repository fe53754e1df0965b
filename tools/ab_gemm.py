#!/usr/bin/env python3
"""Interleaved A/B of the fp32 1x1 (GEMM) variants in one process: python tools/ab_gemm.py "M K N;..." 9,5,13,14 [rounds] [res]"""
import sys, os, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
shapes = [tuple(int(v) for v in s.split()) for s in sys.argv[1].split(';') if s.strip()]
variants = [int(v) for v in sys.argv[2].split(',')]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
use_res = len(sys.argv) > 4 and 'res' in sys.argv[4]
use_gn = len(sys.argv) > 4 and 'gn' in sys.argv[4]
HW = 1024
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
warm = False
for (M, K, N) in shapes:
    torch.manual_seed(0)
    x = torch.randn(M * K, device='cuda'); w0 = torch.randn(N, K, device='cuda') * 0.02
    b = torch.randn(N, device='cuda'); out = torch.empty(M * N, device='cuda'); res = torch.randn(M * N, device='cuda')
    w = torch.empty(lib.nd_conv_weight_floats(N, K, 1), device='cuda')
    assert lib.nd_repack_conv_weight(w0.data_ptr(), w.data_ptr(), N, K, 1, st) == 0
    fl = 2.0 * M * N * K
    gA = torch.randn((M // HW) * K, device='cuda'); gB = torch.randn((M // HW) * K, device='cuda')
    def run(v, n):
        for _ in range(n):
            rc = lib.nd_conv_nhwc(x.data_ptr(), K, K, None, 0, 0, w.data_ptr(), b.data_ptr(), None, 0, res.data_ptr() if use_res else None, N,
                                  out.data_ptr(), N, M // HW, 32, 32, N, 1, 0, v, gA.data_ptr() if use_gn else None, gB.data_ptr() if use_gn else None, K, st)
            if rc != 0: return False
        return True
    ok = [v for v in variants if run(v, 1)]
    if not warm:
        t0 = time.time()
        while time.time() - t0 < 2.0:
            run(ok[0], 10); torch.cuda.synchronize()
        warm = True
    res_t = {v: [] for v in ok}
    outs = {}
    for r in range(rounds):
        for v in ok:
            run(v, 2); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(v, 10); e1.record(); e1.synchronize()
            res_t[v].append(e0.elapsed_time(e1) / 10)
            if r == 0: outs[v] = out.clone()
    ref = outs[ok[0]]
    print((M, K, N), '  '.join('v%d: %.3f ms %.1f TF (maxdiff %.1e)' % (v, statistics.median(res_t[v]), fl / statistics.median(res_t[v]) / 1e9,
                                                                       (outs[v] - ref).abs().max().item()) for v in ok))
