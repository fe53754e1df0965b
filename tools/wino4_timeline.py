#!/usr/bin/env python3
"""Per-block time stamps of conv_wino4_kernel (diagnostic build: tools/build_one_variant.sh qDIAG nd_conv_winograd_quad.hip
-DND_W4_DIAG [-DND_WABL_...]): in-kernel clock, blocks resident per CU over time, prologue / main loop / epilogue spans.
   ND_HIP_LIB=gpurun_variants/libnd_qDIAG.so python tools/wino4_timeline.py NI H W C N [variant]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch, numpy as np
from nicediffusion import _hip
NI, H, W, C, N = [int(v) for v in sys.argv[1:6]]
var = int(sys.argv[6]) if len(sys.argv) > 6 else 12
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.randn(NI * H * W * C, device='cuda'); w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
w = torch.empty(lib.nd_conv_winograd_weight_floats(N, C), device='cuda')
assert lib.nd_repack_conv_weight_winograd(w0.data_ptr(), w.data_ptr(), N, C, st) == 0
b = torch.randn(N, device='cuda'); out = torch.empty(NI * H * W * N, device='cuda')
nblocks = (NI * H * W // 128 + 64) * ((N + 63) // 64) + 64
dbg = torch.zeros(max(nblocks * 8, NI * N), dtype=torch.int32, device='cuda')
def run():
    assert lib.nd_conv3x3_winograd_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), dbg.data_ptr(), N, None, 0,
                                        out.data_ptr(), N, NI, H, W, N, 0, var, None, None, 0, st) == 0, _hip.last_error()
import time
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(10): run()
    torch.cuda.synchronize()
dbg.zero_(); run(); torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.uint32).reshape(-1, 8)
d = d[d[:, 3] != 0]
print('blocks stamped', len(d))
t0s = d[:, 0].astype(np.int64) + (d[:, 7].astype(np.int64) << 32)
base = t0s.min()
start = (t0s - base) / 100.0           # us (100 MHz)
pro, ml_end, end = d[:, 1] / 100.0, d[:, 2] / 100.0, d[:, 3] / 100.0
clk = d[:, 4] / ((d[:, 2] - d[:, 1]).astype(np.float64) * 10.0) * 1e3 / 1e3   # shader cycles per 10 ns -> GHz
print('kernel span %.1f us' % (start + end).max())
print('prologue %.2f us  main loop %.2f us  epilogue %.2f us  (medians); block life %.2f us' % (
    np.median(pro), np.median(ml_end - pro), np.median(end - ml_end), np.median(end)))
print('in-kernel clock GHz: median %.3f  p10 %.3f  p90 %.3f' % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
cu = (d[:, 6].astype(np.int64) << 16) | (d[:, 5] & 0xff00) | ((d[:, 5] >> 13) & 0x7) << 4 | ((d[:, 5] >> 12) & 1)
groups = collections.defaultdict(list)
for i in range(len(d)): groups[cu[i]].append(i)
print('distinct CUs', len(groups), 'blocks per CU min/max', min(len(v) for v in groups.values()), max(len(v) for v in groups.values()))
# residency: for a few CUs print the block intervals
res2 = []
for k, idx in groups.items():
    ev = []
    for i in idx: ev.append((start[i], 1)); ev.append((start[i] + end[i], -1))
    ev.sort(); cur = 0; last = ev[0][0]; t = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    for (tt, dv) in ev:
        t[min(cur, 3)] += tt - last; last = tt; cur += dv
    res2.append((t[0], t[1], t[2], t[3]))
r = np.array(res2); tot = r.sum(1, keepdims=True)
print('share of a CU\'s busy span with 0/1/2/3+ blocks resident: %s' % np.round((r / tot).mean(0), 3))
k0 = sorted(groups.keys())[0]
print('CU', hex(k0), 'timeline (start, prologue, mainloop, epilogue) us, simd/wave of stamping wave:')
for i in sorted(groups[k0], key=lambda i: start[i])[:14]:
    print('   %8.2f  +%.2f  +%.2f  +%.2f   hwid %08x' % (start[i], pro[i], ml_end[i] - pro[i], end[i] - ml_end[i], d[i, 5]))
