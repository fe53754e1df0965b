# Where do the waves of conv_winow_kernel (variant 11) wait?  Separate rocprofv3 --pmc passes; run on the GPU box.
cd /tmp && export TMPDIR=/tmp && export WINO=1
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=${1:-11}
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM -d $R/gpurun_out/pmcw1 -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 32 32 384 384 3 $V 3 > $R/gpurun_out/pmcw1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmcw2 -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 32 32 384 384 3 $V 3 > $R/gpurun_out/pmcw2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmcw3 -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 32 32 384 384 3 $V 3 > $R/gpurun_out/pmcw3.log 2>&1
for i in 1 2 3; do python3 $R/tools/pmc_parse.py $R/gpurun_out/pmcw$i conv_wino >> $R/gpurun_out/pmcw_summary.log 2>&1; done
