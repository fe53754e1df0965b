#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]  -> gpurun_variants/libnd_NAME.so  (A/B experiments; travels to the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=${ND_SRC:-$ROOT/nice-diffusion_amd/csrc}
mkdir -p $ROOT/gpurun_variants /tmp/ndvar_$NAME
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$SRC "$@" -c $f -o /tmp/ndvar_$NAME/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_variants/libnd_$NAME.so /tmp/ndvar_$NAME/*.o
echo built $ROOT/gpurun_variants/libnd_$NAME.so
