#!/bin/bash
# timing-only ablations of conv_wino4_kernel (variant 12); libs built by tools/build_one_variant.sh qNAME nd_conv_winograd_quad.hip -DND_WABL_*
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
LOG=gpurun_out/r3_wino4_abl.log; : > $LOG
for v in "" ${ABL:-qNOEPI qNOB qNOA qNOHALO qNOBAR qSKEL qSKELNOEPI}; do
  for s in "64 64 64 192 192" "64 32 32 384 384"; do
    if [ -z "$v" ]; then L=""; else L=gpurun_variants/libnd_$v.so; fi
    echo "== ${v:-full} $s" >> $LOG
    ND_HIP_LIB=$L WINO=1 timeout -k 10 120 python tools/conv_bench.py $s 3 12 30 2>&1 | grep "shape" >> $LOG || exit 1
  done
done
cat $LOG
