#!/bin/bash
# correctness of the Winograd variants, then variant 8 (16-wave position-split) vs 12 (4-wave, two blocks per CU) on the main shapes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
LOG=gpurun_out/r3_wino4.log; : > $LOG
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "winograd" >> $LOG 2>&1 || { tail -30 $LOG; exit 1; }
for s in "64 64 64 192 192" "64 64 64 384 192" "64 32 32 384 384" "64 32 32 768 384" "64 16 16 576 576" "64 16 16 1152 576" "64 8 8 768 768"; do
  WINO=1 timeout -k 10 120 python tools/conv_bench.py $s 3 8,12 30 2>&1 | grep "shape\|n/a\|diff" >> $LOG || exit 1
done
tail -40 $LOG
