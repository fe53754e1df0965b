#!/usr/bin/env python3
"""Per-shape HBM traffic (and matrix-pipe occupancy) of every convolution launch of a forward, IN the forward:

    python tools/pmc_shapes.py <workload> <fetch_dir> <write_dir> [<sq_dir>]  ->  profiles/r02_pmc_shapes.json (merged)

Inputs are the rocprofv3 output directories of separate --pmc passes over tools/pmc_forward.py (tools/pmc_shapes.sh runs
them): FETCH_SIZE, WRITE_SIZE and optionally GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES.  The
conv dispatches of the last `reps` forwards are matched, in order, with the plan's launch list, and averaged per key
    kind:variant:k<ksize>:NI:H:W:Cin:N
HBM bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 (gfx950 counts 128-byte read requests as 64 bytes:
MI355X_MICROARCH.md, HBM section) + WRITE_SIZE [KiB] x 1024.  bench.py looks its dominant kernel's shapes up here."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV_FNS = ('nd_conv_nhwc', 'nd_conv3x3_winograd_nhwc', 'nd_conv_bf16_nhwc', 'nd_conv3x3_winograd_stats_nhwc',
            'nd_conv3x3_winograd_vstats_nhwc', 'nd_conv3x3_bf16_stats_nhwc', 'nd_conv1x1_bf16_stats_nhwc', 'nd_conv_bf16_splitk_nhwc',
            'nd_conv_splitk_nhwc', 'nd_conv3x3_winograd_splitk_nhwc', 'nd_conv1x1_stats_nhwc', 'nd_conv3x3_winograd_f4_nhwc')
CONV_KERNELS = ('conv_wino16_kernel', 'conv_wino_kernel', 'conv_wino4_kernel', 'conv_wf4_kernel', 'conv_mfma_kernel',
                'gemm_stream_kernel', 'conv_bf16_kernel', 'conv_bf16s_kernel', 'gemm_bf16_kernel', 'gemm_bf16q_kernel', 'gemm_f32_kernel',
                'gemm4_kernel')
# the other kernel classes of a forward: counters aggregated per kernel name (no per-shape key)
CLASS_KERNELS = ('attention_kernel', 'attention_bf16_kernel', 'gn_stats_kernel', 'gn_apply_kernel', 'gn_from_partials_kernel',
                 'gn_coeffs_kernel', 'gn_coeffs_from_partials_kernel', 'gn_fused_small_kernel', 'splitk_reduce_kernel', 'splitk_reduce_f32_kernel')
OUT_NAME = os.environ.get('ND_PMC_OUT', os.environ.get('ROUND', 'r06') + '_pmc_shapes.json')


def dispatches(d, kernels=CONV_KERNELS):
    """[(dispatch id, kernel name, {counter: value}, duration_us)] of the named kernels, in dispatch order."""
    cc = (glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*_counter_collection.csv'))[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        if not any(k in r['Kernel_Name'] for k in kernels) or 'pack_' in r['Kernel_Name']:
            continue
        e = acc.setdefault(int(r['Dispatch_Id']), [r['Kernel_Name'], collections.defaultdict(float), None])
        e[1][r['Counter_Name']] += float(r['Counter_Value'])
    kt = glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv')
    if kt:
        for r in csv.DictReader(open(kt[0])):
            i = int(r['Dispatch_Id'])
            if i in acc:
                acc[i][2] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return [(i,) + tuple(v) for i, v in sorted(acc.items())]


def key_of(m):
    """kind[+stats]:variant:k<ksize>:NI:H:W:Cin:N -- the statistics-writing instantiation of a kernel is a different kernel"""
    kind, var = m['variant'] if m.get('variant') else ('direct', -1)
    if 'stats' in m['fn']:
        kind += '+stats'
    return '{}:{}:k{}:{}'.format(kind, var, m.get('ksize') or 1, ':'.join(str(v) for v in (m.get('shape') or [0, 0, 0, 0, 0])))


def main():
    wl, dirs = sys.argv[1], sys.argv[2:]
    ops = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_ops_{}.json'.format(wl))))
    convs = [m for m in ops['ops'] if m['fn'] in CONV_FNS]
    reps = ops['reps']
    per_key = collections.OrderedDict()
    for d in dirs:
        disp = dispatches(d)
        need = reps * len(convs)
        assert len(disp) >= need, '{}: {} conv dispatches < {} forwards x {} convs'.format(d, len(disp), reps, len(convs))
        disp = disp[-need:]
        for j, (_, kname, ctr, us) in enumerate(disp):
            m = convs[j % len(convs)]
            key = key_of(m)
            e = per_key.setdefault(key, dict(kernel=kname.split('(')[0], label=m['label'], ksize=m.get('ksize'), flops=m['flops'],
                                             n=collections.defaultdict(int), sums=collections.defaultdict(float)))
            # (launches of one shape may run different instantiations of their kernel: with / without a residual)
            assert e['kernel'].split('<')[0] == kname.split('<')[0], 'dispatch order does not match the plan at {}: {} vs {}'.format(
                key, e['kernel'], kname)
            for c, v in ctr.items():
                e['sums'][c] += v
                e['n'][c] += 1
            if us is not None:
                e['sums']['_us'] += us
                e['n']['_us'] += 1
    esize = 2 if ops['dtype'] == 'bf16' else 4

    counts = collections.Counter(key_of(m) for m in convs)
    out = {}
    tot_h = tot_a = 0.0
    for key, e in per_key.items():
        avg = {c: e['sums'][c] / e['n'][c] for c in e['sums']}
        rec = dict(kernel=e['kernel'], label=e['label'], launches_per_forward=counts[key],
                   duration_us=round(avg.get('_us', 0.0), 1), flops=e['flops'])
        NI, H, W, Cin, N = [int(v) for v in key.split(':')[3:]]
        k = e['ksize'] or 1
        rec['algorithmic_bytes'] = esize * (NI * H * W * (Cin + N) + k * k * Cin * N)
        if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg and rec['algorithmic_bytes'] > 0:
            rec['fetch_kib_raw'] = round(avg['FETCH_SIZE'], 1)
            rec['write_kib'] = round(avg['WRITE_SIZE'], 1)
            rec['hbm_bytes'] = int(avg['FETCH_SIZE'] * 1024 * 2 + avg['WRITE_SIZE'] * 1024)
            rec['traffic_ratio'] = round(rec['hbm_bytes'] / rec['algorithmic_bytes'], 3)
        if 'GRBM_GUI_ACTIVE' in avg and avg['GRBM_GUI_ACTIVE'] > 0:
            cyc = avg['GRBM_GUI_ACTIVE'] / 8
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
                rec['mfma_busy'] = round(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc, 4)      # 256 CUs x 4 SIMDs
            if avg.get('_us'):
                rec['clock_ghz'] = round(cyc / avg['_us'] / 1e3, 3)
        if avg.get('SQ_WAVE_CYCLES'):
            rec['wait_any'] = round(avg.get('SQ_WAIT_ANY', 0.0) / avg['SQ_WAVE_CYCLES'], 4)
        out[key] = rec
    # ---- the non-conv classes (attention, GroupNorm): per kernel name, the dispatches of the LAST forward's share of the run
    classes = {}
    for d in dirs:
        disp = dispatches(d, CLASS_KERNELS)
        if not disp:
            continue
        byname = collections.defaultdict(list)
        for (_, kname, ctr, us) in disp:
            byname[kname.split('(')[0].replace('void ', '')].append((ctr, us))
        for kname, lst in byname.items():
            # keep the profiled forwards only: the last reps/(warm + reps) share is not known per kernel, so average over all
            e = classes.setdefault(kname, dict(n=collections.defaultdict(int), sums=collections.defaultdict(float)))
            for ctr, us in lst:
                for c, v in ctr.items():
                    e['sums'][c] += v
                    e['n'][c] += 1
                if us is not None:
                    e['sums']['_us'] += us
                    e['n']['_us'] += 1
    cls_out = {}
    nops = collections.Counter()
    for m in ops['ops']:
        nops[m['fn']] += 1
    for kname, e in classes.items():
        avg = {c: e['sums'][c] / e['n'][c] for c in e['sums']}
        rec = dict(avg_duration_us=round(avg.get('_us', 0.0), 2), dispatches_seen=max(e['n'].values()))
        if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg:
            rec['hbm_bytes_per_launch'] = int(avg['FETCH_SIZE'] * 1024 * 2 + avg['WRITE_SIZE'] * 1024)
            rec['fetch_kib_raw'] = round(avg['FETCH_SIZE'], 1)
            rec['write_kib'] = round(avg['WRITE_SIZE'], 1)
            if avg.get('_us'):
                rec['hbm_gbps'] = round(rec['hbm_bytes_per_launch'] / avg['_us'] / 1e3, 1)
        if avg.get('GRBM_GUI_ACTIVE', 0) > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
            rec['mfma_busy'] = round(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (avg['GRBM_GUI_ACTIVE'] / 8), 4)
        if avg.get('SQ_WAVE_CYCLES'):
            rec['wait_any'] = round(avg.get('SQ_WAIT_ANY', 0.0) / avg['SQ_WAVE_CYCLES'], 4)
        cls_out[kname] = rec
    path = os.path.join(ROOT, 'profiles', OUT_NAME)
    merged = {}
    if os.path.exists(path):
        merged = json.load(open(path))
    merged.update(out)
    # per-kernel totals over one forward of this workload
    by_kernel = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for key, rec in out.items():
        if 'hbm_bytes' in rec:
            n = counts[key]
            by_kernel[rec['kernel']][0] += rec['hbm_bytes'] * n
            by_kernel[rec['kernel']][1] += rec['algorithmic_bytes'] * n
            by_kernel[rec['kernel']][2] += n
    merged['_totals_' + wl] = {k: dict(hbm_bytes_per_forward=int(v[0]), algorithmic_bytes_per_forward=int(v[1]),
                                       ratio=round(v[0] / v[1], 3), launches=v[2]) for k, v in by_kernel.items()}
    merged['_classes_' + wl] = cls_out
    # which library build the counters belong to (bench.py flags a table taken on another build as stale)
    merged.setdefault('_stamps', {})[wl] = ops.get('stamp')
    merged['_comment'] = ('Generated by tools/pmc_shapes.sh (rocprofv3 --pmc passes over tools/pmc_forward.py, separate passes for '
                          'FETCH_SIZE / WRITE_SIZE / SQ counters) and tools/pmc_shapes.py; key = kind:variant:k<ksize>:NI:H:W:Cin:N of a '
                          'conv launch IN the forward; hbm_bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x '
                          '1024 per launch; algorithmic_bytes = esize x (input + output elements + weights).')
    tmp = path + '.tmp'          # never leave a half-written table behind (bench.py reads it)
    json.dump(merged, open(tmp, 'w'), indent=1, sort_keys=True)
    os.replace(tmp, path)
    for k, v in merged['_totals_' + wl].items():
        print(k, v)
    for k, v in cls_out.items():
        print(k, v)


if __name__ == '__main__':
    main()
