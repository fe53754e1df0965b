#!/usr/bin/env python3
"""Per-wave time stamps of conv_bf16_kernel (diagnostic build: tools/build_one_variant.sh hDIAG nd_conv_bf16.hip -DND_BF_DIAG
[-DND_HABL_...]): in-kernel clock, blocks resident per CU, prologue / per-chunk / epilogue spans.
   ND_HIP_LIB=gpurun_variants/libnd_hDIAG.so python tools/bf16_timeline.py NI H W C N [variant] [plain|stats|gn|gnstats]
Without a DIAG build the same script only times the launch (the stamps buffer stays zero)."""
import sys, os, collections, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch, numpy as np
from nicediffusion import _hip
NI, H, W, C, N = [int(v) for v in sys.argv[1:6]]
var = int(sys.argv[6]) if len(sys.argv) > 6 else 11
mode = sys.argv[7] if len(sys.argv) > 7 else 'stats'
lib = _hip.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.randn(NI * H * W * C, device='cuda').to(torch.bfloat16)
w0 = torch.randn(N, C, 3, 3, device='cuda') * 0.02
w = torch.empty(lib.nd_conv_bf16_weight_elems(N, C, 3), dtype=torch.bfloat16, device='cuda')
assert lib.nd_repack_conv_weight_bf16(w0.data_ptr(), w.data_ptr(), N, C, 3, lib.nd_conv_bf16_variant_layout(var), st) == 0
b = torch.randn(N, device='cuda'); out = torch.empty(NI * H * W * N, dtype=torch.bfloat16, device='cuda')
bm, bn, nthr = (ctypes.c_int() for _ in range(3))
lib.nd_conv_bf16_variant_info(var, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nthr))
waves = nthr.value // 64
nblocks = (NI * H * W // bm.value + 64) * ((N + bn.value - 1) // bn.value) + 64
dbg = torch.zeros(max(nblocks * waves * 16, NI * N), dtype=torch.int32, device='cuda')
gnA = gnB = None
if mode.startswith('gn'):
    gnA = torch.rand(NI, C, device='cuda') + 0.5; gnB = torch.randn(NI, C, device='cuda') * 0.1
rows = lib.nd_conv_bf16_stats_rows(NI, H, W, N, var) if mode.endswith('stats') else 0
cs = torch.empty(NI * max(rows, 1) * 2 * N, device='cuda') if rows > 0 else None
flags = _hip.CONV_GN_SILU if gnA is not None else 0
def run(stamp):
    rb = dbg.data_ptr() if stamp else None
    if rows > 0:
        rc = lib.nd_conv3x3_bf16_stats_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), rb, N, None, 0,
                                            out.data_ptr(), N, NI, H, W, N, flags, var, _hip.ptr(gnA), _hip.ptr(gnB), C,
                                            cs.data_ptr(), st)
    else:
        rc = lib.nd_conv_bf16_nhwc(x.data_ptr(), C, C, None, 0, 0, w.data_ptr(), b.data_ptr(), rb, N, None, 0,
                                   out.data_ptr(), N, NI, H, W, N, 3, flags, var, _hip.ptr(gnA), _hip.ptr(gnB), C, st)
    assert rc == 0, _hip.last_error()
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(10): run(False)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run(False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * NI * H * W * N * 9 * C
print('shape', (NI, H, W, C, N), 'variant', var, mode, '%.4f ms  %.0f TFLOP/s' % (ms, fl / ms / 1e9))
dbg.zero_(); run(True); torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.uint32).reshape(-1, 16)
d = d[d[:, 3] != 0]
if len(d) == 0:
    print('no stamps (not a DIAG build)'); sys.exit(0)
print('waves stamped', len(d))
t0s = d[:, 0].astype(np.int64) + (d[:, 7].astype(np.int64) << 32)
base = t0s.min()
start = (t0s - base) / 100.0           # us (100 MHz)
pro, ml_end, end = d[:, 1] / 100.0, d[:, 2] / 100.0, d[:, 3] / 100.0
clk = d[:, 4] / ((d[:, 2] - d[:, 1]).astype(np.float64) * 10.0)
print('kernel span %.1f us' % (start + end).max())
print('prologue %.2f us  main loop %.2f us  epilogue %.2f us  (medians); wave life %.2f us' % (
    np.median(pro), np.median(ml_end - pro), np.median(end - ml_end), np.median(end)))
print('   p10/p90: prologue %.2f/%.2f  main %.2f/%.2f  epilogue %.2f/%.2f' % (
    np.percentile(pro, 10), np.percentile(pro, 90), np.percentile(ml_end - pro, 10), np.percentile(ml_end - pro, 90),
    np.percentile(end - ml_end, 10), np.percentile(end - ml_end, 90)))
nch = (C + 63) // 64
chs = d[:, 8:8 + min(nch, 8)] / 100.0
prev = pro
line = []
for i in range(chs.shape[1]):
    line.append('%.2f' % np.median(chs[:, i] - prev)); prev = chs[:, i]
print('chunk spans (median us):', ' '.join(line))
if nch <= 4:
    e = d[:, 13:16] / 100.0
    print('epilogue stamps after main loop (median us): first pixel row stored +%.2f  all stored +%.2f  drained +%.2f' % (
        np.median(e[:, 0] - ml_end), np.median(e[:, 1] - ml_end), np.median(np.where(e[:, 2] > 0, e[:, 2] - ml_end, 0))))
print('in-kernel clock GHz: median %.3f  p10 %.3f  p90 %.3f' % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
mfma_cyc = nch * 9 * 4 * (bm.value // 32) * (bn.value // 32) / waves * 32
print('MFMA cycles per wave %d = %.2f us at the median clock; x2 waves per SIMD = %.2f us' % (
    mfma_cyc, mfma_cyc / np.median(clk) / 1e3, 2 * mfma_cyc / np.median(clk) / 1e3))
cu = (d[:, 6].astype(np.int64) << 16) | (d[:, 5] & 0xff00) | ((d[:, 5] >> 13) & 0x7) << 4 | ((d[:, 5] >> 12) & 1)
groups = collections.defaultdict(list)
for i in range(len(d)): groups[cu[i]].append(i)
print('distinct CUs', len(groups), 'waves per CU min/max', min(len(v) for v in groups.values()), max(len(v) for v in groups.values()))
res2 = []
for k, idx in groups.items():
    ev = []
    for i in idx: ev.append((start[i], 1)); ev.append((start[i] + end[i], -1))
    ev.sort(); cur = 0; last = ev[0][0]; t = collections.defaultdict(float)
    for (tt, dv) in ev:
        t[min(cur, 9)] += tt - last; last = tt; cur += dv
    res2.append([t[j] for j in range(10)])
r = np.array(res2); tot = r.sum(1, keepdims=True)
print('share of a CU\'s busy span with 0..9 waves resident: %s' % np.round((r / tot).mean(0), 3))
# phase occupancy: share of the CU's wave-time in prologue / main / epilogue
wt = end.sum(); print('wave-time shares: prologue %.3f  main %.3f  epilogue %.3f' % (pro.sum() / wt, (ml_end - pro).sum() / wt, (end - ml_end).sum() / wt))
k0 = sorted(groups.keys())[0]
print('CU', hex(k0), 'timeline of its waves (start, +prologue, +mainloop, +epilogue) us:')
for i in sorted(groups[k0], key=lambda i: start[i])[:24]:
    print('   %8.2f  +%.2f  +%.2f  +%.2f   hwid %08x  chunks %s' % (start[i], pro[i], ml_end[i] - pro[i], end[i] - ml_end[i], d[i, 5],
                                                               ' '.join('%.1f' % v for v in chs[i])))
