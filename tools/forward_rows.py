#!/usr/bin/env python3
"""Per-launch table of one eager UNet forward of a bench workload (HIP-event times on the launch stream):
    python tools/forward_rows.py config4 [min_ms]
One line per launch: time, entry point, label, shape / variant where the plan recorded them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

wl_name = sys.argv[1] if len(sys.argv) > 1 else 'config2'
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
wl = bench.WORKLOADS[wl_name]
dev = torch.device('cuda:0')
margs, model, diff = bench.build(dev, wl)
NI = wl['batch'] * (2 if wl['cfg'] is not None else 1)
plan, rows = bench.kernel_breakdown(model, NI)
tot = sum(r['ms'] for r in rows)
print('%d launches, %.3f ms' % (len(rows), tot))
for r in rows:
    if r['ms'] >= min_ms:
        print('%8.4f  %-34s %-28s %s %s' % (r['ms'], r['fn'], r['label'], r.get('shape') or '', r.get('variant') or ''))
