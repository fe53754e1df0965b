# Round profile set (run on the GPU box from the repo root): per-shape PMC tables (conv launches keyed by shape + the
# attention / GroupNorm classes) for config2 / config4, and rocprofv3 --kernel-trace --stats summaries of bench.py for the
# three workloads (eager launches, short chains).  The PMC table is rebuilt in a scratch file and moved over the
# committed one only when every pass succeeded (tools/pmc_shapes.py).
set -u
ROUND=${ROUND:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
for WL in ${PMC_WORKLOADS:-config2 config4}; do
  bash $R/tools/pmc_shapes.sh $WL > $R/gpurun_out/pmc_sh_$WL.log 2>&1 || echo "pmc $WL failed"
done
cd /tmp && export TMPDIR=/tmp
for WL in ${STATS_WORKLOADS:-config2 config4 config5}; do
  CH=10; [ $WL = config5 ] && CH=5
  rm -rf $R/gpurun_out/stats_$WL
  ND_TUNE_CACHE=$R/gpurun_out/tune_$WL.json timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$WL -o run --output-format csv -- \
      python3 $R/bench.py --workload $WL --chain $CH --no-graph --no-cpu-baseline > $R/gpurun_out/stats_$WL.json 2> $R/gpurun_out/stats_$WL.err || echo "stats $WL failed"
done
# the box only returns gpurun_out/: leave copies of what belongs under profiles/ there
cp $R/profiles/${ROUND}_pmc_shapes.json $R/gpurun_out/${ROUND}_pmc_shapes.json 2>/dev/null
for WL in ${STATS_WORKLOADS:-config2 config4 config5}; do
  cp $R/gpurun_out/stats_$WL/run_kernel_stats.csv $R/gpurun_out/${ROUND}_kernel_stats_${WL}_chain_eager.csv 2>/dev/null
  cp $R/gpurun_out/stats_$WL.json $R/gpurun_out/${ROUND}_bench_under_rocprof_${WL}_eager.json 2>/dev/null
  cp $R/gpurun_out/tune_$WL.json $R/gpurun_out/${ROUND}_tune_cache_$WL.json 2>/dev/null
done
ls $R/gpurun_out/stats_config2 | head
