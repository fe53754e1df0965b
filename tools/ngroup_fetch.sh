# FETCH_SIZE (HBM / fabric read bytes, x2 on gfx950) and run time of a Winograd variant under different block orders.
# Run on the GPU box:  bash tools/ngroup_fetch.sh [variant=12]   -> gpurun_out/ngf.txt
set -u
V=${1:-12}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ngf.txt
cd /tmp && export TMPDIR=/tmp && export WINO=1
for s in "64 64 64 192 192" "64 32 32 384 384" "64 16 16 576 576" "64 64 64 384 384"; do
  for g in 1 2 3 4 6 99; do
    export ND_NGROUP=$g
    D=$R/gpurun_out/ngf_${g}
    rm -rf $D
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D -o runc --output-format csv -- python3 $R/tools/conv_bench.py $s 3 $V 3 > $R/gpurun_out/ngf_run.log 2>&1 || exit 1
    echo "shape $s ngroup $g: $(python3 $R/tools/pmc_parse.py $D conv_wino | tail -1)" >> $R/gpurun_out/ngf.txt
  done
done
cat $R/gpurun_out/ngf.txt
