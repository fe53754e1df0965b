mkdir -p gpurun_out
BENCH_WORKLOADS="" STATS_WORKLOADS="" ROUND=r05 bash tools/profile_round.sh > gpurun_out/profile_round_b.log 2>&1
tail -8 gpurun_out/profile_round_b.log
cp profiles/r05_pmc_shapes.json gpurun_out/r05_pmc_shapes.json
