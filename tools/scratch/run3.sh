set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "first or last_conv or gather" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 64 > gpurun_out/ops_config2_edge.txt 2>&1
grep "first\|taps\|gather" gpurun_out/ops_config2_edge.txt
