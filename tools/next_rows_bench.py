#!/usr/bin/env python3
"""Measurement of the HBM-bound kernels either side of the hot path (SURVEY 8(f) rows N1-N3 and the K10 sampler
update) against the HBM roofline, plus the one-time weight ingest.  GPU box only:

    python tools/next_rows_bench.py [images]        (default 4096 images of 3x64x64 -> 201 MB per fp32 NHWC4 tensor)

Prints one JSON line per kernel: algorithmic bytes, time (HIP events, 20 launches), GB/s and fraction of 8 TB/s."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
import torch
from nicediffusion import _hip, default_args as DA
from nicediffusion.model import DiffusionModel
from nicediffusion.diffusion import Diffusion

PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md
NI = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R, C, CP = 64, 3, 4
HW = R * R
lib = _hip.load()
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, nbytes, ms, note=''):
    gbs = nbytes / ms / 1e6
    print(json.dumps({'kernel': name, 'algorithmic_bytes': nbytes, 'ms': round(ms, 4), 'GB/s': round(gbs, 1),
                      'frac_of_8TBps': round(gbs / PEAK, 3), 'note': note}))


torch.manual_seed(0)
x_nchw = torch.randn(NI, C, R, R, device=dev)
x_nhwc = torch.empty(NI * HW * CP, device=dev)
eps = torch.randn(NI * HW * 8, device=dev)           # 6-channel model output padded to 8
noise = torch.randn(NI * HW * CP, device=dev)
out = torch.empty_like(x_nhwc)
u8 = torch.empty(NI * HW * C, dtype=torch.uint8, device=dev)

ck = _hip.check
report('nd_nchw_to_nhwc', NI * HW * (C + CP) * 4,
       timed(lambda: ck(lib.nd_nchw_to_nhwc(x_nchw.data_ptr(), x_nhwc.data_ptr(), NI, C, HW, CP, st))),
       'edge of denoise: read NCHW 3ch, write NHWC padded to 4')
report('nd_nhwc_to_nchw', NI * HW * (C + CP) * 4,
       timed(lambda: ck(lib.nd_nhwc_to_nchw(x_nhwc.data_ptr(), x_nchw.data_ptr(), NI, C, HW, CP, st))))
report('nd_to_uint8_hwc (N2)', NI * HW * (CP * 4 + C),
       timed(lambda: ck(lib.nd_to_uint8_hwc(x_nhwc.data_ptr(), CP, u8.data_ptr(), NI, HW, C, 0, st))),
       'sample.py:94-100: read fp32 NHWC4, write uint8 HWC')
report('nd_qsample (N3)', NI * HW * CP * 4 * 3,
       timed(lambda: ck(lib.nd_qsample(x_nhwc.data_ptr(), noise.data_ptr(), out.data_ptr(), x_nhwc.numel(), 0.8, 0.6, st))),
       'diffusion.py:232-240: 2 reads + 1 write')

# sampler update on the real coefficient table (K10)
m = DiffusionModel(**DA.EMNIST_MODEL_ARGS).to(dev)    # any model: only the schedule tables are used here
d = Diffusion(m, 1000, 250, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=True, ddim_eta=0.0, device=dev)
coef = d.coefficient_table().to(dev)
step = torch.full((1,), 100, dtype=torch.int32, device=dev)
report('nd_ddim_step (K10, eta=0)', NI * HW * (CP * 4 * 2 + 8 * 4),
       timed(lambda: ck(lib.nd_ddim_step(x_nhwc.data_ptr(), out.data_ptr(), None, None, 0, CP, eps.data_ptr(), None, 8, 0.0,
                                         coef.data_ptr(), step.data_ptr(), 0.0, None, 0, 0, None, 0, NI, HW, C, st))),
       'read x (NHWC4) + model output (NHWC8), write x')
report('nd_ddpm_step (K10, learned_interpolation, Philox noise)', NI * HW * (CP * 4 * 2 + 8 * 4),
       timed(lambda: ck(lib.nd_ddpm_step(x_nhwc.data_ptr(), out.data_ptr(), None, None, 0, CP, eps.data_ptr(), None, 8, 0.0,
                                         coef.data_ptr(), step.data_ptr(), 2, None, 0, 1234, None, 0, NI, HW, C, st))),
       'same traffic; noise generated in-kernel')

# N1: weight ingest of the 64x64 preset: load_state_dict + fragment-order repack of every conv / linear (plan build
# without autotuning)
os.environ['ND_AUTOTUNE'] = '0'
t0 = time.time()
big = DiffusionModel(**DA.OPENAI_64_MODEL_ARGS)
sd = big.state_dict()
t1 = time.time()
big.load_state_dict(sd, strict=True)
big = big.to(dev)
torch.cuda.synchronize()
t2 = time.time()
plan = big._plan(1)
torch.cuda.synchronize()
t3 = time.time()
nparam = sum(p.numel() for p in big.parameters())
print(json.dumps({'step': 'weight ingest, 64x64 preset (N1)', 'params': nparam,
                  'construct_s': round(t1 - t0, 2), 'load_state_dict_plus_H2D_s': round(t2 - t1, 2),
                  'plan_build_and_repack_s (no autotune, B=1)': round(t3 - t2, 2),
                  'packed_weight_floats': int(plan.packed_floats)}))
