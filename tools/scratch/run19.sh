set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
for i in 1 2 3; do
for v in attKT128 attKT64; do
echo $v; ND_HIP_LIB=gpurun_variants/libnd_$v.so python tools/attn_bench.py 64 1024 6 64 20 2>&1 | grep attention
done; done
