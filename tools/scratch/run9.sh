set -e
mkdir -p gpurun_out
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 32 config4 > gpurun_out/ops_config4.txt 2>&1
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 16 config5 > gpurun_out/ops_config5.txt 2>&1
head -45 gpurun_out/ops_config4.txt
