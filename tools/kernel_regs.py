#!/usr/bin/env python3
"""Print VGPR / AGPR / scratch use per kernel from hipcc -S output (amdhsa metadata).  usage: kernel_regs.py file.s [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
meta = txt[txt.rfind('amdhsa.kernels:'):]
for blk in meta.split('\n  - ')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    name = name.replace('nd::', '').replace('(nd::ConvArgs)', '').replace('void ', '')
    if flt in name:
        print('{:60s} vgpr {:>4} agpr {:>4} scratch {:>5} lds {:>6}'.format(name[:60], g('vgpr_count'), g('agpr_count'),
              g('private_segment_fixed_size'), g('group_segment_fixed_size')))
