# PMC passes (separate, as the microarchitecture guide prescribes) for the dominant Winograd kernel; run on the GPU box
cd /tmp && export TMPDIR=/tmp && export WINO=1
R=$GRAFT_REPO_ROOT
V=${1:-6}
i=0
for s in "64 16 16 576 576" "64 64 64 384 192"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmcd_sq$i -o runc --output-format csv -- python3 $R/tools/conv_bench.py $s 3 $V 3 > $R/gpurun_out/pmcd.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcd_fetch$i -o runc --output-format csv -- python3 $R/tools/conv_bench.py $s 3 $V 3 >> $R/gpurun_out/pmcd.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmcd_write$i -o runc --output-format csv -- python3 $R/tools/conv_bench.py $s 3 $V 3 >> $R/gpurun_out/pmcd.log 2>&1 || exit 1
done
