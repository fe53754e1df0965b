#!/usr/bin/env python3
"""Micro-benchmark of nd_conv_bf16_nhwc on one shape, all tile variants (GPU box only):
   python tools/conv_bench_bf16.py NI H W Cin N ksize [variants(comma)|all] [iters]
Random bf16 data (zero operands would let the chip clock higher: MI355X_MICROARCH.md, DVFS give-back)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip
a = sys.argv[1:]
NI, H, W, C, N, ks = [int(v) for v in a[:6]]
lib = _hip.load()
nv = lib.nd_conv_bf16_num_variants()
variants = list(range(nv)) if len(a) <= 6 or a[6] == 'all' else [int(v) for v in a[6].split(',')]
iters = int(a[7]) if len(a) > 7 else 20
dev = 'cuda'
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(NI * H * W * C, device=dev).to(torch.bfloat16)
w0 = torch.randn(N, C, ks, ks, device=dev) * 0.02
ws = []
for lay in (0, 1):
    wl = torch.empty(lib.nd_conv_bf16_weight_elems(N, C, ks), dtype=torch.bfloat16, device=dev)
    assert lib.nd_repack_conv_weight_bf16(w0.data_ptr(), wl.data_ptr(), N, C, ks, lay, st) == 0
    ws.append(wl)
b = torch.randn(N, device=dev)
out = torch.empty(NI * H * W * N, dtype=torch.bfloat16, device=dev)
fl = 2.0 * NI * H * W * N * ks * ks * C
warm = False
for v in variants:
    w = ws[lib.nd_conv_bf16_variant_layout(v)]

    def run():
        return lib.nd_conv_bf16_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0,
                                     out.data_ptr(), N, NI, H, W, N, ks, 0, v, None, None, 0, st)
    if run() != 0:
        print('variant', v, 'n/a:', _hip.last_error())
        continue
    torch.cuda.synchronize()
    if not warm:
        t0 = time.time()
        while time.time() - t0 < 1.5:
            for _ in range(10):
                run()
            torch.cuda.synchronize()
        warm = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bm, bn, nt = (__import__('ctypes').c_int() for _ in range(3))
    lib.nd_conv_bf16_variant_info(v, *(__import__('ctypes').byref(z) for z in (bm, bn, nt)))
    print('shape', (NI, H, W, C, N, ks), 'variant %d (%dx%d, %d thr)' % (v, bm.value, bn.value, nt.value),
          '%.3f ms  %.0f TFLOP/s (%.0f %% of 2500)' % (ms, fl / ms / 1e9, fl / ms / 1e9 / 25))
