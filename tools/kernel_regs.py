#!/usr/bin/env python3
"""VGPR / AGPR / scratch / spill counts per kernel, read from the amdhsa metadata of the BUILT library (or an object file).

    kernel_regs.py [libnd_hip.so | file.o] [filter]
    kernel_regs.py libnd_hip.so conv_wf4_kernel loops      # loops of the machine code with their VMEM / LDS / MFMA counts

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


def _code_objects(path, tmp):
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat])
    data = open(fat, 'rb').read()
    starts = [m.start() for m in re.finditer(MAGIC, data)]
    for i, a in enumerate(starts):
        piece = os.path.join(tmp, 'c{}.bin'.format(i))
        open(piece, 'wb').write(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = os.path.join(tmp, 'c{}.co'.format(i))
        subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + piece,
                               '--targets=' + TARGET, '--output=' + co], stderr=subprocess.DEVNULL)
        yield co


_COUNTED = ('buffer_load', 'global_load', 'scratch_load', 'flat_load', 'buffer_store', 'global_store', 'scratch_store', 'flat_store',
            'buffer_atomic', 'global_atomic', 'v_mfma', 'ds_read', 'ds_write', 's_load', 's_buffer_load', 's_barrier', 's_waitcnt')


def loop_table(path, name_filter):
    """{demangled kernel name: [loop, ...]} for the kernels whose mangled name contains ``name_filter``: every loop of the
    machine code (a backward branch and its target) with its instruction count and the number of instructions per opcode
    family that matter to a hand-counted ``s_waitcnt``: VMEM loads (LDS-DMA loads = ``buffer_load_*`` with the ``lds``
    modifier are kept apart as ``<opcode>_lds``), stores, atomics, scalar loads (they count on lgkmcnt), LDS reads / writes,
    MFMAs, barriers.  The hand-scheduled kernels issue their run-ahead loads as inline ISA and wait with literal counts: a
    compiler that adds ONE load to such a loop (a rematerialised argument, a reloaded descriptor) shifts every count."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(path, tmp):
            syms = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-s', '-W', co], capture_output=True, text=True).stdout
            if name_filter not in syms:
                continue
            text = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', co], capture_output=True, text=True).stdout
            cur, kern = None, {}
            for ln in text.split('\n'):
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
                if m:
                    cur = m.group(1) if name_filter in m.group(1) else None
                    if cur:
                        kern[cur] = []
                    continue
                if cur:
                    m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):', ln)
                    if m:
                        kern[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
            for name, ins in kern.items():
                idx = {a: i for i, (a, _, _) in enumerate(ins)}
                loops = []
                for i, (a, op, args) in enumerate(ins):
                    if not (op.startswith('s_cbranch') or op == 's_branch'):
                        continue
                    try:
                        simm = int(args.split()[0])
                    except (ValueError, IndexError):
                        continue
                    if simm >= 32768:
                        simm -= 65536
                    tgt = a + 4 + 4 * simm
                    if tgt <= a and tgt in idx:
                        cnt = {}
                        for _, o, ar in ins[idx[tgt]:i + 1]:
                            if o.startswith(_COUNTED):
                                k = o + '_lds' if (o.startswith('buffer_load') and re.search(r'\blds\b', ar)) else o
                                cnt[k] = cnt.get(k, 0) + 1
                        loops.append(dict(head=tgt, tail=a, instructions=i + 1 - idx[tgt], counts=cnt))
                dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
                out[dem] = loops
    return out


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'nice-diffusion_amd', 'nicediffusion', 'libnd_hip.so')
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    if len(sys.argv) > 3 and sys.argv[3] == 'loops':
        for name, loops in sorted(loop_table(path, flt).items()):
            print(name)
            for lp in loops:
                print('   loop {:#x}..{:#x} {:5d} instructions  {}'.format(lp['head'], lp['tail'], lp['instructions'], lp['counts']))
        sys.exit(0)
    for name, r in sorted(kernel_table(path).items()):
        short = name.replace('nd::', '').replace('(nd::ConvArgs)', '').replace('void ', '')
        if flt in short:
            print('{:72s} vgpr {:>4} agpr {:>4} scratch {:>5} lds {:>6} spills s{} v{}'.format(
                short[:72], r['vgpr'], r['agpr'], r['scratch'], r['lds'], r['sgpr_spill'], r['vgpr_spill']))
