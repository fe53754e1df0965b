export WINO=1
for s in "64 32 32 384 384" "64 64 64 192 192"; do
  for lib in "" nob nohalo noa noall; do
    L=""; [ -n "$lib" ] && L=gpurun_variants/libnd_$lib.so
    echo "lib=$lib"; ND_HIP_LIB=$L timeout -k 10 120 python tools/conv_bench.py $s 3 8 20 2>&1 | grep -E "^shape|n/a"
  done
done
