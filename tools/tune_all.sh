set -u
mkdir -p gpurun_out
for WL in config2 config4 config5 config1; do
  python bench.py --workload $WL --steps 2 --warmup 1 --retune --save-tune-cache gpurun_out/tune_cache_$WL.json > gpurun_out/r05_pre_bench_$WL.json 2> gpurun_out/r05_pre_bench_$WL.err || echo "bench $WL failed"
  tail -1 gpurun_out/r05_pre_bench_$WL.err
done
