#!/bin/bash
# tools/bf16_timeline.sh: per-wave timelines and timing-only ablations of the dominant bf16 conv on its main shapes
# (libraries from tools/build_one_variant.sh hDIAG* nd_conv_bf16.hip -DND_BF_DIAG [-DND_HABL_*])
set -u
O=gpurun_out/bf16_timeline.log; : > $O
for s in "32 128 128 256 256" "32 128 128 512 256"; do
  for m in plain stats gn gnstats; do
    echo "=== full $s $m" >> $O
    ND_HIP_LIB=gpurun_variants/libnd_hDIAG.so timeout -k 10 120 python tools/bf16_timeline.py $s 11 $m >> $O 2>&1 || exit 1
  done
done
for a in NOHALO NOB NOA NOEPI NOBAR; do
  for m in stats gnstats; do
    echo "=== $a 32 128 128 256 256 $m" >> $O
    ND_HIP_LIB=gpurun_variants/libnd_hDIAG$a.so timeout -k 10 120 python tools/bf16_timeline.py 32 128 128 256 256 11 $m 2>&1 | grep -v "^   \|^CU" >> $O || exit 1
  done
done
