for s in "64 64 64 192" "64 32 32 384" "64 16 16 576" "64 8 8 768" "64 32 32 768"; do
  for lib in gpurun_variants/libnd_gnold.so ""; do
    echo "shape $s lib=$lib"; ND_HIP_LIB=$lib timeout -k 10 120 python tools/gn_bench.py $s 20 2>&1 | grep -v amdgpu
  done
done
