for s in "64 32 32 384 1152 1" "64 16 16 576 1728 1" "64 64 64 384 192 1" "64 32 32 384 384 1" "64 64 64 384 384 3" "64 15 15 384 384 3"; do
  for lib in "" gpurun_variants/libnd_inter.so; do
    echo "lib=$lib"; ND_HIP_LIB=$lib timeout -k 10 120 python tools/conv_bench.py $s 1,5,9,12 20 2>&1 | grep -E "shape|n/a"
  done
done
