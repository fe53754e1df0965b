#!/usr/bin/env python3
"""VGPR / AGPR / scratch / spill counts per kernel, read from the amdhsa metadata of the BUILT library (or an object file).

    kernel_regs.py [libnd_hip.so | file.o] [filter]

``kernel_table(path)`` -> {demangled kernel name: dict(vgpr, agpr, scratch, lds, sgpr_spill, vgpr_spill)}.  The .hip_fatbin
section holds one clang offload bundle per translation unit; each is unbundled for gfx950 and its notes are parsed."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def kernel_table(path):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat])
        data = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(MAGIC, data)]
        mangled = {}
        for i, a in enumerate(starts):
            piece = os.path.join(tmp, 'b{}.bin'.format(i))
            open(piece, 'wb').write(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
            co = os.path.join(tmp, 'b{}.co'.format(i))
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + piece,
                                   '--targets=' + TARGET, '--output=' + co], stderr=subprocess.DEVNULL)
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True).stdout
            meta = notes[notes.find('amdhsa.kernels:'):]
            for blk in re.split(r'\n\s+- (?=\.)', meta)[1:]:
                g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, None])[1]
                if g('name') is None or g('vgpr_count') is None:
                    continue
                mangled[g('name')] = dict(vgpr=int(g('vgpr_count')), agpr=int(g('agpr_count') or 0),
                                          scratch=int(g('private_segment_fixed_size') or 0),
                                          lds=int(g('group_segment_fixed_size') or 0),
                                          sgpr_spill=int(g('sgpr_spill_count') or 0), vgpr_spill=int(g('vgpr_spill_count') or 0))
        names = list(mangled)
        try:
            dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
        except OSError:
            dem = names
        for n, d in zip(names, dem):
            out[d.strip() or n] = mangled[n]
    return out


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'nice-diffusion_amd', 'nicediffusion', 'libnd_hip.so')
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, r in sorted(kernel_table(path).items()):
        short = name.replace('nd::', '').replace('(nd::ConvArgs)', '').replace('void ', '')
        if flt in short:
            print('{:72s} vgpr {:>4} agpr {:>4} scratch {:>5} lds {:>6} spills s{} v{}'.format(
                short[:72], r['vgpr'], r['agpr'], r['scratch'], r['lds'], r['sgpr_spill'], r['vgpr_spill']))
