timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r2_t19.log 2>&1 || exit 1
for s in "64 32 32 384 1152" "64 32 32 384 384" "64 64 64 576 192" "64 64 64 384 192" "64 16 16 576 1728" "64 16 16 576 576" "64 32 32 768 384" "64 8 8 768 2304" "64 16 16 1344 576" "64 32 32 192 384"; do
  timeout -k 10 120 python tools/conv_bench.py $s 1 0,1,5,9,11,12,13 20 2>&1 | grep shape >> gpurun_out/r2_g32.log
done
