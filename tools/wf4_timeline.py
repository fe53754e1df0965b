#!/usr/bin/env python3
"""Per-workgroup time stamps of conv_wf4_kernel (diagnostic build: tools/build_one_variant.sh f4DIAG nd_conv_winograd_f4.hip
-falign-loops=64 -DND_F4_DIAG): prologue / main loop / exchange / tail spans of wave 0, and -- by CU (HW_ID, XCC_ID) -- the gap
between a workgroup's end and the start of the next one on the same CU.
   ND_HIP_LIB=gpurun_variants/libnd_f4DIAG.so python tools/wf4_timeline.py NI H W C N [res]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch, numpy as np
from nicediffusion import _hip
NI, H, W, C, N = [int(v) for v in sys.argv[1:6]]
want_res = len(sys.argv) > 6 and sys.argv[6] == 'res'
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.randn(NI * H * W * C, device='cuda'); w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
w = torch.empty(lib.nd_conv_winograd_f4_weight_floats(0, N, C), device='cuda')
assert lib.nd_repack_conv_weight_winograd_f4(w0.data_ptr(), w.data_ptr(), N, C, 0, st) == 0
b = torch.randn(N, device='cuda'); out = torch.empty(NI * H * W * N, device='cuda')
res = torch.randn(NI * H * W * N, device='cuda') if want_res else None
rows = lib.nd_conv_winograd_f4_stats_rows(0, NI, H, W)
stats = torch.empty(NI * rows * 2 * N, device='cuda')
dbg = torch.zeros(max(8 * 65536, NI * N), dtype=torch.int32, device='cuda')


def run():
    assert lib.nd_conv3x3_winograd_f4_nhwc(x.data_ptr(), C, C, w.data_ptr(), b.data_ptr(), dbg.data_ptr(), N, None if res is None else res.data_ptr(),
                                           0 if res is None else N, out.data_ptr(), N, NI, H, W, N, 0, 0, stats.data_ptr(), 1, None, st) == 0, _hip.last_error()


import time
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(10):
        run()
    torch.cuda.synchronize()
dbg.zero_(); run(); torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.uint32).reshape(-1, 8)
d = d[d[:, 4] != 0]
print('workgroups stamped', len(d))
t0s = d[:, 0].astype(np.int64) + (d[:, 7].astype(np.int64) << 32)
base = t0s.min()
start = (t0s - base) / 100.0           # us (100 MHz)
ta = (d[:, 1] >> 16) / 100.0          # entry -> first DMA about to be issued (arguments, descriptors, addresses)
tb = (d[:, 6] >> 16) / 100.0          # ... -> both chunks and the first fragments requested
d[:, 1] &= 0xffff
t1, t2, t3, t4 = (d[:, k] / 100.0 for k in (1, 2, 3, 4))
print('kernel span %.1f us' % (start + t4).max())
med = np.median
print('prologue in parts (medians): setup %.2f us | issue of 2 chunks + 6 fragments %.2f us | wait + barrier %.2f us' % (med(ta), med(tb - ta), med(t1 - tb)))
print('medians: entry -> first barrier (prologue) %.2f us | main loop %.2f us | loop end -> exchange barrier %.2f us | units (reads, transform, '
      'store issue) %.2f us | workgroup life (wave 0) %.2f us' % (med(t1), med(t2 - t1), med(t3 - t2), med(t4 - t3), med(t4)))
# by CU: (xcc, se, sh, cu) from HW_ID bits: cu_id [11:8], sh_id [12], se_id [15:13]
hw = d[:, 5]
cu = ((d[:, 6] & 0xf).astype(np.int64) << 16) | (((hw >> 13) & 7).astype(np.int64) << 8) | (((hw >> 12) & 1).astype(np.int64) << 4) | ((hw >> 8) & 0xf)
gaps, per_cu = [], collections.Counter()
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    order = idx[np.argsort(start[idx])]
    per_cu[len(order)] += 1
    for a, bb in zip(order[:-1], order[1:]):
        gaps.append(start[bb] - (start[a] + t4[a]))
gaps = np.array(gaps)
print('distinct CUs %d, workgroups per CU %s' % (len(np.unique(cu)), dict(per_cu)))
if len(gaps):
    print('gap between the end of a workgroup (wave 0 past its store issue) and the entry of the next one on the same CU: median %.2f us, p10 %.2f, '
          'p90 %.2f' % (med(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90)))
rounds = len(d) / max(1, len(np.unique(cu)))
print('per round of workgroups: life %.2f + gap %.2f = %.2f us; x %.1f rounds = %.1f us' % (med(t4), med(gaps) if len(gaps) else 0, med(t4) + (med(gaps) if len(gaps) else 0),
                                                                                       rounds, rounds * (med(t4) + (med(gaps) if len(gaps) else 0))))
# spread of start times inside a round (are the CUs in phase?)
first = np.sort(start)[:len(np.unique(cu))]
print('first round: entries within %.2f us (p10-p90 %.2f)' % (first.max() - first.min(), np.percentile(first, 90) - np.percentile(first, 10)))
