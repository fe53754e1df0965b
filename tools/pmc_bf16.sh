# SQ counters of conv_bf16_kernel (variant 11, 128 px x 256 ch, two blocks per CU) on its headline shape, launched alone:
#   bash tools/pmc_bf16.sh     (on the GPU box; three separate --pmc passes, --kernel-trace only)
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "b:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAIT_INST_LDS" "c:SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  rm -rf $R/gpurun_out/pmc_bf16_$NAME
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS -d $R/gpurun_out/pmc_bf16_$NAME -o run --output-format csv -- \
      python3 $R/tools/bf16_timeline.py 32 128 128 256 256 11 stats > $R/gpurun_out/pmc_bf16_$NAME.log 2>&1 || { tail -5 $R/gpurun_out/pmc_bf16_$NAME.log; exit 1; }
done
cd $R && python3 - <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for name in 'abc':
    f = glob.glob('gpurun_out/pmc_bf16_%s/**/*counter_collection.csv' % name, recursive=True)
    if not f:
        print('no csv for pass', name); continue
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'conv_bf16_kernel' not in k: continue
        key = (k.split('(')[0][:70], r['Grid_Size'])
        d = out.setdefault(key, collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in out.items():
    print(key, 'launches counted', len(d['SQ_WAVE_CYCLES']))
    wc = sum(d['SQ_WAVE_CYCLES']) / max(1, len(d['SQ_WAVE_CYCLES']))
    for c, v in d.items():
        m = sum(v) / len(v)
        print('   %-32s %14.0f  %6.3f of SQ_WAVE_CYCLES' % (c, m, m / wc if wc else 0))
PY
