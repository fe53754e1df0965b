mkdir -p gpurun_out
BENCH_WORKLOADS="config2 config4 config5" PMC_WORKLOADS="" STATS_WORKLOADS="config2" ROUND=r05 bash tools/profile_round.sh > gpurun_out/profile_round_c.log 2>&1
tail -6 gpurun_out/profile_round_c.log
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 64 > gpurun_out/r05_layer_table_config2.txt 2>&1
