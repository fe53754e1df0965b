# timing-only ablations of conv_wino16_kernel (variant 8): full | no epilogue | skeleton (no operand loads) | skeleton without epilogue
for v in "" wgNOEPI wgSKEL wgSKELNOEPI; do
  for s in "64 64 64 192 192" "64 32 32 384 384" "64 16 16 576 576"; do
    if [ -z "$v" ]; then L=""; else L=gpurun_variants/libnd_$v.so; fi
    echo "== ${v:-full} $s" >> gpurun_out/r2_w16abl.log
    ND_HIP_LIB=$L WINO=1 timeout -k 10 120 python tools/conv_bench.py $s 3 8 30 2>&1 | grep shape >> gpurun_out/r2_w16abl.log
  done
done
