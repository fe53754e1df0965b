#!/usr/bin/env python3
"""Micro-benchmark of nd_attention_nhwc: python tools/attn_bench.py B T heads hd [iters]   (ND_ATTN_WAVES=4|8 forces the block size)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
B, T, heads, hd = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
C = heads * hd
qkv = torch.randn(B * T * 3 * C, device='cuda'); out = torch.empty(B * T * C, device='cuda')
def run():
    rc = lib.nd_attention_nhwc(qkv.data_ptr(), 3 * C, out.data_ptr(), C, B, T, heads, hd, 0, C, 2 * C, hd, hd ** -0.5, st)
    assert rc == 0, _hip.last_error()
run(); run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 4.0 * B * heads * T * T * hd
print('attention B=%d T=%d heads=%d hd=%d waves=%s: %.4f ms  %.1f TFLOP/s' % (B, T, heads, hd, os.environ.get('ND_ATTN_WAVES', 'auto'), ms, fl / ms / 1e9))
