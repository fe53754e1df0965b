#!/bin/bash
# tools/build_wave_variant.sh NAME [flags]: libnd_NAME.so = the current build with nd_conv_winograd_wave.hip recompiled with the flags
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/gpurun_variants /tmp/ndw_$NAME
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$ROOT/nice-diffusion_amd/csrc "$@" -c $ROOT/nice-diffusion_amd/csrc/nd_conv_winograd_wave.hip -o /tmp/ndw_$NAME/w.o
OBJS=$(ls $ROOT/nice-diffusion_amd/build/*.o | grep -v nd_conv_winograd_wave.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_variants/libnd_$NAME.so $OBJS /tmp/ndw_$NAME/w.o
echo built libnd_$NAME.so
