# Round profile set (run on the GPU box from the repo root): per-shape PMC tables for config2 / config4, and
# rocprofv3 --kernel-trace --stats summaries of bench.py for the three workloads (eager launches, short chains).
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/profiles/r02_pmc_shapes.json
bash $R/tools/pmc_shapes.sh config2 > $R/gpurun_out/pmc_sh_config2.log 2>&1 || echo "pmc config2 failed"
bash $R/tools/pmc_shapes.sh config4 > $R/gpurun_out/pmc_sh_config4.log 2>&1 || echo "pmc config4 failed"
cd /tmp && export TMPDIR=/tmp
for WL in config2 config4 config5; do
  CH=10; [ $WL = config5 ] && CH=5
  rm -rf $R/gpurun_out/stats_$WL
  ND_TUNE_CACHE=$R/gpurun_out/tune_$WL.json timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$WL -o run --output-format csv -- \
      python3 $R/bench.py --workload $WL --chain $CH --no-graph --no-cpu-baseline > $R/gpurun_out/stats_$WL.json 2> $R/gpurun_out/stats_$WL.err || echo "stats $WL failed"
done
ls $R/gpurun_out/stats_config2 | head
