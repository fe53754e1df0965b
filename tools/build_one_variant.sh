#!/bin/bash
# tools/build_one_variant.sh NAME FILE.hip [flags]: gpurun_variants/libnd_NAME.so = the current build with csrc/FILE.hip
# recompiled with the flags (A/B experiments and timing-only ablations; travels to the GPU box, select with ND_HIP_LIB).
# Also `make -C nice-diffusion_amd variant NAME=.. FILE=.. FLAGS=..`.  A library built with any macro listed in
# csrc/nd_variant_flags.inc reports it through nd_build_flags() and loads only under ND_ALLOW_ABLATION=1; the flags are
# part of the stamp of everything measured with it.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; shift; shift
mkdir -p $ROOT/gpurun_variants /tmp/ndv_$NAME
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$ROOT/nice-diffusion_amd/csrc "$@" -c $ROOT/nice-diffusion_amd/csrc/$FILE -o /tmp/ndv_$NAME/v.o
OBJS=$(ls $ROOT/nice-diffusion_amd/build/*.o | grep -v "/${FILE%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_variants/libnd_$NAME.so $OBJS /tmp/ndv_$NAME/v.o
echo built libnd_$NAME.so
