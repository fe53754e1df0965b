mkdir -p gpurun_out
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_config2_driver_flags.json 2> gpurun_out/r05_bench_config2_driver_flags.err
rc=$?
T1=$(date +%s)
echo "wall_s $((T1-T0))" >> gpurun_out/r05_bench_config2_driver_flags.err
tail -2 gpurun_out/r05_bench_config2_driver_flags.err
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 64 > gpurun_out/r05_layer_table_config2.txt 2>&1
exit $rc
