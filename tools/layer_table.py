#!/usr/bin/env python3
"""Per-launch table of one eager UNet forward (HIP events): shape, tile variant, ms, TFLOP/s.  GPU box only.
    python tools/layer_table.py [forward batch] [workload]      (ND_LAYER_TABLE_OPS=1: every launch in plan order)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd')); sys.path.insert(0, ROOT)
import torch
import bench
CONV = bench.CONV_FNS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
WL = sys.argv[2] if len(sys.argv) > 2 else 'config2'          # forward batch B (2x the image batch under guidance)
from nicediffusion import _engine
_engine.preload_tune_cache(os.path.join(ROOT, 'profiles', 'tune_cache_%s.json' % WL), override=True)
margs, model, diff = bench.build(torch.device('cuda'), bench.WORKLOADS[WL])
plan, rows = bench.kernel_breakdown(model, B, reps=3)
tot = sum(r['ms'] for r in rows)
print('forward %.2f ms, %.1f TFLOP/s' % (tot, plan.flops / tot / 1e9))
agg = {}
for r in rows:
    if r['fn'] in CONV and r['shape']:
        k = (r['ksize'],) + tuple(r['shape']) + (r['variant'],)
        a = agg.setdefault(k, [0, 0.0, 0])
        a[0] += 1; a[1] += r['ms']; a[2] += r['flops']
print('%-4s %-26s %-3s %5s %9s %8s %7s' % ('k', 'NI,H,W,Cin,N', 'var', 'calls', 'ms_total', 'ms_avg', 'TF/s'))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-4d %-26s %-10s %5d %9.3f %8.3f %7.1f' % (k[0], ','.join(map(str, k[1:6])), '%s%d' % (k[6][0][:4], k[6][1]), a[0], a[1], a[1] / a[0], a[2] / a[1] / 1e9))
oth = {}
for r in rows:
    if r['fn'] not in CONV or not r['shape']:
        oth[r['fn']] = oth.get(r['fn'], 0) + r['ms']
print({k: round(v, 3) for k, v in oth.items()})
if os.environ.get('ND_LAYER_TABLE_OPS'):
    # every launch in plan order: index, entry point, label, ms
    for i, r in enumerate(rows):
        print('%4d %-40s %-44s %8.4f %s' % (i, r['fn'], r['label'], r['ms'], '' if not r['shape'] else ','.join(map(str, r['shape']))))
