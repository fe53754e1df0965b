#!/usr/bin/env python3
"""DESIGN.md = tools/DESIGN.tmpl.md with the @@PLACEHOLDER@@ numbers of section 5 filled from the bench lines under
profiles/ (run after the final benches of a round): python tools/fill_design.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    return json.loads(open(os.path.join(ROOT, 'profiles', name)).read().strip().splitlines()[-1])


c2 = line('r02_bench_config2_250step.json')
c4 = line('r02_bench_config4_1000step_bf16.json')
c5 = line('r02_bench_config5_50step_bf16.json')
v = {}
for k, d in (('C2', c2), ('C4', c4), ('C5', c5)):
    v[k + '_VALUE'] = '**{:.3f}**'.format(d['value']) if k != 'C5' else '**{:.2f}**'.format(d['value'])
    v[k + '_MS'] = '{:.2f}'.format(d['ms_per_sampler_step'])
    v[k + '_CPU'] = '{:.2g}'.format(d.get('cpu_baseline', {}).get('value', float('nan')))
for k, d in (('C2', c2), ('C4', c4)):
    r, f = d['roofline'], d['forward']
    c = f['ms_by_class']
    v[k + '_ACH'] = '{:.1f}'.format(r['achieved'])
    v[k + '_FRAC'] = '**{:.3f}**'.format(r['frac'])
    v[k + '_ALG'] = '{:.1f} TFLOP/s'.format(r['algorithmic_equivalent'])
    v[k + '_AVG'] = '{:.4f}'.format(r['avg_launch_ms'])
    v[k + '_C3'] = '{:.1f}'.format(c.get('conv3x3', 0))
    v[k + '_C1'] = '{:.1f}'.format(c.get('conv1x1', 0))
    v[k + '_AT'] = '{:.1f}'.format(c.get('attention', 0))
    v[k + '_GS'] = '{:.2f}'.format(c.get('groupnorm_stats', 0))
    v[k + '_GA'] = '{:.2f}'.format(c.get('groupnorm_apply', 0))
    v[k + '_GC'] = '{:.2f}'.format(c.get('groupnorm_coeffs', 0))
    v[k + '_FWD'] = '{:.1f}'.format(f['eager_sum_of_kernels_ms'])
    v[k + '_FWDFRAC'] = '{:.3f}'.format(f['executed_frac_of_matrix_peak'])
v['C4_MSIMG'] = '{:.2f}'.format(c4['forward']['eager_sum_of_kernels_ms'] / c4['forward']['images'])
p = os.path.join(ROOT, 'DESIGN.md')
s = open(os.path.join(ROOT, 'tools', 'DESIGN.tmpl.md')).read()      # the template holds the placeholders; edit IT, not DESIGN.md
missing = set(re.findall(r'@@(\w+)@@', s)) - set(v)
assert not missing, missing
for k, val in v.items():
    s = s.replace('@@' + k + '@@', val)
open(p, 'w').write(s)
print('filled', len(v), 'values')
