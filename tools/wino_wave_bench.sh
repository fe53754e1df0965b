set -e
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" > gpurun_out/r2_t17.log 2>&1
for s in "64 64 64 192 192" "64 32 32 384 384" "64 16 16 576 576" "64 8 8 768 768" "64 32 32 768 384" "64 64 64 384 192"; do
  WINO=1 timeout -k 10 120 python tools/conv_bench.py $s 3 8,11 30 >> gpurun_out/r2_wbench.log 2>&1
done
