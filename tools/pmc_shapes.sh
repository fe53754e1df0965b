# Regenerates profiles/${ROUND:-r06}_pmc_shapes.json (ND_PMC_OUT names another file): per-shape HBM traffic + matrix-pipe counters of every conv launch inside a
# forward.  Run on the GPU box from the repo root:   bash tools/pmc_shapes.sh [config2|config4|config5]
# (separate --pmc passes with --kernel-trace only, the python program directly after `--`).
WL=${1:-config2}
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
export ND_TUNE_CACHE=$R/gpurun_out/tune_$WL.json
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/pmc_forward.py $WL 1 > $R/gpurun_out/pmc_$WL.log 2>&1 || exit 1          # unprofiled: fills the tune cache
for P in fetch:FETCH_SIZE write:WRITE_SIZE "sq:GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  rm -rf $R/gpurun_out/pmc_${WL}_$NAME
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $CTRS -d $R/gpurun_out/pmc_${WL}_$NAME -o run --output-format csv -- \
      python3 $R/tools/pmc_forward.py $WL 2 >> $R/gpurun_out/pmc_$WL.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_shapes.py $WL gpurun_out/pmc_${WL}_fetch gpurun_out/pmc_${WL}_write gpurun_out/pmc_${WL}_sq
