# FETCH_SIZE (HBM/fabric read bytes) of the Winograd kernel under different block orders; run on the GPU box
cd /tmp && export TMPDIR=/tmp && export WINO=1
R=$GRAFT_REPO_ROOT
for g in 1 2 4 99; do
  export ND_NGROUP=$g
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/ngf_$g -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 64 64 192 192 3 5 2 > $R/gpurun_out/ngf_$g.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/ngf_b$g -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 32 32 384 384 3 5 2 >> $R/gpurun_out/ngf_$g.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/ngf_c$g -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 16 16 576 576 3 5 2 >> $R/gpurun_out/ngf_$g.log 2>&1 || exit 1
done
