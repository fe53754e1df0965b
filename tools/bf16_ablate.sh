#!/bin/bash
# tools/bf16_ablate.sh [build|run]: the per-phase table of conv_bf16_kernel<1,4,4,2,9,...> (variant 11: 128 px x 256 ch, two blocks
# per CU) on the two shapes VERDICT r5 names -- 128x128 256->256 at NI = 32 (configs[3]) and 64x64 512->512 -- from per-wave stamps
# (-DND_BF_DIAG: in-kernel clock = s_memtime / s_memrealtime, prologue / chunk / epilogue spans) and timing-only ablation builds
# (-DND_HABL_*; wrong results by construction, loaded under ND_ALLOW_ABLATION=1):
#   BASE      the kernel as shipped                      SKEL      NOTHING but the MFMAs of this tile (+ loop control): the ceiling
#   NOEPI     no epilogue                                ONLYB     SKEL + the weight-fragment loads (global -> VGPR)
#   NOB       no weight-fragment loads                   ONLYA     SKEL + the LDS fragment reads
#   NOA       no LDS fragment reads                      ONLYHALO  SKEL + the halo fetches and the chunk barrier
#   NOHALO    only chunk 0 is fetched                    ONLYEPI   SKEL + the epilogue
#   NOBAR     no chunk barrier
#   build: here (CPU, hipcc cross-compiles); run: on the GPU box -> gpurun_out/r06_bf16_ablations.log
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
ALL="NOHALO NOB NOA NOEPI NOBAR"
declare -A V
V[BASE]=""
V[NOEPI]="-DND_HABL_NOEPI"; V[NOB]="-DND_HABL_NOB"; V[NOA]="-DND_HABL_NOA"; V[NOHALO]="-DND_HABL_NOHALO"; V[NOBAR]="-DND_HABL_NOBAR"
V[SKEL]="-DND_HABL_NOEPI -DND_HABL_NOB -DND_HABL_NOA -DND_HABL_NOHALO -DND_HABL_NOBAR"
V[ONLYB]="-DND_HABL_NOEPI -DND_HABL_NOA -DND_HABL_NOHALO -DND_HABL_NOBAR"
V[ONLYA]="-DND_HABL_NOEPI -DND_HABL_NOB -DND_HABL_NOHALO -DND_HABL_NOBAR"
V[ONLYHALO]="-DND_HABL_NOEPI -DND_HABL_NOB -DND_HABL_NOA"
V[ONLYEPI]="-DND_HABL_NOB -DND_HABL_NOA -DND_HABL_NOHALO -DND_HABL_NOBAR"
ORDER="BASE NOEPI NOB NOA NOHALO NOBAR SKEL ONLYB ONLYA ONLYHALO ONLYEPI"
if [ "${1:-run}" = build ]; then
  n=0
  for k in $ORDER; do
    bash $R/tools/build_one_variant.sh hD$k nd_conv_bf16.hip -DND_BF_DIAG ${V[$k]} > /tmp/bf16abl_$k.log 2>&1 &
    n=$((n+1)); [ $((n % 4)) = 0 ] && wait
  done
  wait; ls -la $R/gpurun_variants/libnd_hD*.so; exit 0
fi
O=$R/gpurun_out/r06_bf16_ablations.log; : > $O
export ND_ALLOW_ABLATION=1
for s in "32 128 128 256 256" "32 64 64 512 512"; do
  for m in stats plain; do
    for k in $ORDER; do
      echo "=== $k $s $m" >> $O
      ND_HIP_LIB=$R/gpurun_variants/libnd_hD$k.so timeout -k 10 120 python3 $R/tools/bf16_timeline.py $s 11 $m 2>&1 | grep -v "^   \|^CU\|amdgpu.ids" >> $O || exit 1
    done
    echo "done $s $m"
  done
done
