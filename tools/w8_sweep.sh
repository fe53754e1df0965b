for s in "64 1024 6 64" "64 256 9 64"; do
  for lib in "" anoload ""; do
    L=""; [ -n "$lib" ] && L=gpurun_variants/libnd_$lib.so
    ND_HIP_LIB=$L timeout -k 10 120 python tools/attn_bench.py $s 200 2>&1 | grep attention | sed "s/^/lib=$lib /"
  done
done
