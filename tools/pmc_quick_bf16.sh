cd /tmp && export TMPDIR=/tmp
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for V in 0 12; do
  rm -rf $R/gpurun_out/pmcq_$V
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmcq_$V -o run --output-format csv -- python3 $R/tools/conv_bench_bf16.py 32 64 64 512 512 3 $V 5 > $R/gpurun_out/pmcq_$V.log 2>&1
done
cd $R && for V in 0 12; do python3 tools/pmc_parse.py gpurun_out/pmcq_$V conv_bf16; done
