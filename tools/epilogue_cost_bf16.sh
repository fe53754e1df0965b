# what the epilogue of the dominant bf16 conv costs: full kernel | ideally coalesced stores (wrong values) | no epilogue
for s in "32 128 128 256 256" "16 256 256 256 256"; do
  for v in "" hEPICOAL hNOEPI; do
    echo "== ${v:-full} $s" >> gpurun_out/r2_epi.log
    if [ -z "$v" ]; then L=""; else L=gpurun_variants/libnd_$v.so; fi
    ND_HIP_LIB=$L timeout -k 10 120 python tools/conv_bench_bf16.py $s 3 11 30 2>&1 | grep -i "variant" >> gpurun_out/r2_epi.log
  done
done
