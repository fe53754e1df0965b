#!/usr/bin/env python3
"""Micro-benchmark of nd_attention_bf16_nhwc: python tools/attn_bench_bf16.py B T heads hd [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
B, T, heads, hd = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
C = heads * hd
qkv = torch.randn(B * T * 3 * C, device='cuda').to(torch.bfloat16); out = torch.empty(B * T * C, dtype=torch.bfloat16, device='cuda')
def run():
    rc = lib.nd_attention_bf16_nhwc(qkv.data_ptr(), 3 * C, out.data_ptr(), C, B, T, heads, hd, 0, C, 2 * C, hd, hd ** -0.5, st)
    assert rc == 0, _hip.last_error()
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 4.0 * B * heads * T * T * hd
print('attention bf16 B=%d T=%d heads=%d hd=%d: %.4f ms  %.1f TFLOP/s' % (B, T, heads, hd, ms, fl / ms / 1e9))
