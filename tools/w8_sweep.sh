for s in "64 32 32 384 1152 1" "64 64 64 384 192 1"; do
  for lib in "" nob nohalo noa noall noepi; do
    L=""; [ -n "$lib" ] && L=gpurun_variants/libnd_$lib.so
    echo "lib=$lib"; ND_HIP_LIB=$L timeout -k 10 120 python tools/conv_bench.py $s 1,5 20 2>&1 | grep -E "^shape|n/a"
  done
done
