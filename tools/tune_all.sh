# Regenerate the committed kernel choices after ANY change under csrc/ (their stamp carries nd_build_id, the hash of the
# sources): run on the GPU box, then copy gpurun_out/tune_cache_<workload>.json over profiles/tune_cache_<workload>.json.
# Short chains: the choices depend on the layer shapes only.
set -u
mkdir -p gpurun_out
for WL in config2 config4 config5 config1; do
  python3 bench.py --workload $WL --steps 1 --warmup 1 --chain 10 --no-cpu-baseline --no-breakdown --retune \
      --save-tune-cache gpurun_out/tune_cache_$WL.json > gpurun_out/tune_bench_$WL.json 2> gpurun_out/tune_bench_$WL.err || echo "tune $WL failed"
  tail -1 gpurun_out/tune_bench_$WL.err
  echo "tuned $WL"
done
# the fp32 plans of configs[3] / [4] at their full forward batch (tests/test_gpu_model.py::test_full_batch_fp32_forward_rows_vs_reference
# checks them against the reference's rows): the test itself writes the cache under ND_TUNE_CACHE
for WL in config4 config5; do
  rm -f gpurun_out/tune_cache_${WL}_fp32.json
  ND_TUNE_CACHE=gpurun_out/tune_cache_${WL}_fp32.json python3 -m pytest tests/test_gpu_model.py -m gpu -x -q \
      -k "test_full_batch_fp32_forward_rows_vs_reference and $WL" > gpurun_out/tune_fp32_$WL.log 2>&1 || echo "tune fp32 $WL failed"
  echo "tuned ${WL}_fp32"
done
