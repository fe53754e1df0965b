#!/usr/bin/env python3
"""Program profiled by tools/pmc_shapes.sh: builds the UNet plan of a bench.py workload (tile choices from ND_TUNE_CACHE,
so no tuning launches happen under the profiler), loads the GPU for ~1.5 s so the clock has settled, then runs REPS eager
forwards and writes the plan's launch list (entry point, kernel kind / variant, shape, algorithmic flops) to
gpurun_out/pmc_ops_<workload>.json.  tools/pmc_shapes.py zips that list with the profiler's per-dispatch counters.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 tools/pmc_forward.py config2 3
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else 'config2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = bench.WORKLOADS[wl_name]
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
margs, model, diff = bench.build(dev, wl)
NI = wl['batch'] * (2 if wl['cfg'] is not None else 1)
plan = model._plan(NI)
R = margs['resolution']
torch.manual_seed(0)
plan.x_in.copy_(torch.randn(plan.x_in.numel()).to(dev))
plan.t_in.fill_(500)
if plan.y_in is not None:
    plan.y_in.copy_(((torch.arange(NI) * 37) % 1000).to(dev))
t0 = time.time()
while time.time() - t0 < 1.5:
    plan.run()
    torch.cuda.synchronize()
out = os.path.join(ROOT, 'gpurun_out', 'pmc_ops_{}.json'.format(wl_name))
os.makedirs(os.path.dirname(out), exist_ok=True)
from nicediffusion import _engine  # noqa: E402
json.dump({'workload': wl_name, 'NI': NI, 'reps': reps, 'dtype': wl['dtype'], 'ops': plan.meta, 'stamp': _engine._tune_stamp()}, open(out, 'w'))
torch.cuda.synchronize()
for _ in range(reps):
    plan.run()
torch.cuda.synchronize()
print('pmc_forward: {} launches per forward, {} forwards'.format(len(plan.meta), reps))
