# Round profile set (run on the GPU box from the repo root; ROUND=r06 by default):
#   1. bench lines of the four workloads (the committed tune caches profiles/tune_cache_<workload>.json are loaded by bench.py);
#   2. rocprofv3 --kernel-trace --stats summaries of bench.py for config2 / config4 / config5 (eager launches, short chains);
#   3. per-shape PMC tables (conv launches keyed by shape + the attention / GroupNorm classes) for config2 / config4 / config5.
# Everything lands in gpurun_out/${ROUND}_* (the box only returns gpurun_out/); copy what is to be judged into profiles/.
set -u
ROUND=${ROUND:-r06}
export ROUND
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
for WL in ${BENCH_WORKLOADS-config2 config4 config5 config1}; do
  N=3; [ $WL = config1 ] && N=5
  python3 bench.py --workload $WL --steps $N --warmup 1 > gpurun_out/${ROUND}_bench_$WL.json 2> gpurun_out/${ROUND}_bench_$WL.err || echo "bench $WL failed"
  echo "bench $WL done"
done
for WL in ${PMC_WORKLOADS-config2 config4 config5}; do
  cp $R/profiles/tune_cache_$WL.json $R/gpurun_out/tune_$WL.json
  bash $R/tools/pmc_shapes.sh $WL > $R/gpurun_out/pmc_sh_$WL.log 2>&1 || echo "pmc $WL failed"
  echo "pmc $WL done"
done
cd /tmp && export TMPDIR=/tmp
for WL in ${STATS_WORKLOADS-config2 config4 config5}; do
  CH=10; [ $WL = config5 ] && CH=5
  rm -rf $R/gpurun_out/stats_$WL
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$WL -o run --output-format csv -- \
      python3 $R/bench.py --workload $WL --chain $CH --no-graph --no-cpu-baseline > $R/gpurun_out/stats_$WL.json 2> $R/gpurun_out/stats_$WL.err || echo "stats $WL failed"
  cp $R/gpurun_out/stats_$WL/run_kernel_stats.csv $R/gpurun_out/${ROUND}_kernel_stats_${WL}_chain_eager.csv 2>/dev/null
  cp $R/gpurun_out/stats_$WL.json $R/gpurun_out/${ROUND}_bench_under_rocprof_${WL}_eager.json 2>/dev/null
  echo "stats $WL done"
done
cp $R/profiles/${ROUND}_pmc_shapes.json $R/gpurun_out/${ROUND}_pmc_shapes.json 2>/dev/null
ls $R/gpurun_out/stats_config2 | head
