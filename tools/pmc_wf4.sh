# SQ counters of conv_wf4_kernel (and conv_wino4_kernel beside it) on the headline shapes: bash tools/pmc_wf4.sh  (on the GPU box)
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "b:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU" "c:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  rm -rf $R/gpurun_out/pmc_wf4_$NAME
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS -d $R/gpurun_out/pmc_wf4_$NAME -o run --output-format csv -- \
      python3 $R/tools/wf4_check.py time > $R/gpurun_out/pmc_wf4_$NAME.log 2>&1 || { tail -5 $R/gpurun_out/pmc_wf4_$NAME.log; exit 1; }
done
cd $R && python3 - <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for name in 'abc':
    f = glob.glob('gpurun_out/pmc_wf4_%s/**/*counter_collection.csv' % name, recursive=True)
    if not f:
        print('no csv for pass', name); continue
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'conv_wf4' not in k and 'conv_wino4' not in k: continue
        key = (k.split('(')[0][:60], r['Grid_Size'])
        d = out.setdefault(key, collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in out.items():
    print(key)
    wc = sum(d['SQ_WAVE_CYCLES']) / max(1, len(d['SQ_WAVE_CYCLES']))
    for c, v in d.items():
        m = sum(v) / len(v)
        print('   %-32s %14.0f  %6.3f of SQ_WAVE_CYCLES' % (c, m, m / wc if wc else 0))
PY
