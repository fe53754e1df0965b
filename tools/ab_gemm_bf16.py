#!/usr/bin/env python3
"""Interleaved A/B of the bf16 1x1 forms on one shape (GPU box only):
   python tools/ab_gemm_bf16.py NI H W C N [variants, default 20,21] [res]
Rounds of 20 launches per variant, alternating, 6 rounds; median per variant."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch, numpy as np
from nicediffusion import _hip
NI, H, W, C, N = [int(v) for v in sys.argv[1:6]]
variants = [int(v) for v in sys.argv[6].split(',')] if len(sys.argv) > 6 else [20, 21]
res = len(sys.argv) > 7 and sys.argv[7] == 'res'
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.randn(NI * H * W * C, device='cuda').to(torch.bfloat16)
w0 = torch.randn(N, C, device='cuda') * 0.02
ws = {}
for lay in (0, 1):
    wl = torch.empty(lib.nd_conv_bf16_weight_elems(N, C, 1), dtype=torch.bfloat16, device='cuda')
    assert lib.nd_repack_conv_weight_bf16(w0.data_ptr(), wl.data_ptr(), N, C, 1, lay, st) == 0
    ws[lay] = wl
b = torch.randn(N, device='cuda')
r = torch.randn(NI * H * W * N, device='cuda').to(torch.bfloat16) if res else None
outs = {v: torch.empty(NI * H * W * N, dtype=torch.bfloat16, device='cuda') for v in variants}
def run(v):
    return lib.nd_conv_bf16_nhwc(x.data_ptr(), C, C, None, 0, 0, ws[lib.nd_conv_bf16_variant_layout(v)].data_ptr(), b.data_ptr(), None, 0,
                                 _hip.ptr(r), N if res else 0, outs[v].data_ptr(), N, NI, H, W, N, 1, 0, v, None, None, 0, st)
ok = []
for v in variants:
    if run(v) != 0: print('variant', v, 'n/a:', _hip.last_error())
    else: ok.append(v)
torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 1.5:
    for v in ok: run(v)
    torch.cuda.synchronize()
ts = {v: [] for v in ok}
for rnd in range(6):
    for v in ok:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(v)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 20)
fl = 2.0 * NI * H * W * N * C
by = (NI * H * W * (C + N) + N * C) * 2 + (NI * H * W * N * 2 if res else 0)
for v in ok:
    ms = float(np.median(ts[v]))
    print('shape', (NI, H, W, C, N), 'res' if res else '', 'variant %2d  %.4f ms  %5.0f TFLOP/s  %.2f TB/s algorithmic' % (v, ms, fl / ms / 1e9, by / ms / 1e9))
if len(ok) > 1:
    print('   bitwise equal:', all(torch.equal(outs[ok[0]], outs[v]) for v in ok[1:]))
