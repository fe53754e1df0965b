# timing-only ablations of conv_winow_kernel (results are wrong by construction): skeleton (wALL) + one operand stream at a time
for v in wbase wPK wALL wonlyW wonlyRAW wonlyHALO wonlyXF wonlyBAR; do
  for s in "64 32 32 384 384"; do
    echo "== $v $s" >> gpurun_out/r2_wabl.log
    ND_HIP_LIB=gpurun_variants/libnd_$v.so WINO=1 timeout -k 10 120 python tools/conv_bench.py $s 3 11 30 2>&1 | grep shape >> gpurun_out/r2_wabl.log
  done
done
