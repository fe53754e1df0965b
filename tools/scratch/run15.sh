set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "groupnorm" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
python tools/layer_table.py 64 2>&1 | tail -1 > gpurun_out/gn_rows_on.txt
ND_GN_APPLY_ROWS=0 python tools/layer_table.py 64 2>&1 | tail -1 > gpurun_out/gn_rows_off.txt
python tools/layer_table.py 64 2>&1 | tail -1 >> gpurun_out/gn_rows_on.txt
cat gpurun_out/gn_rows_on.txt gpurun_out/gn_rows_off.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py -x -q -m gpu -k "tiny_forward or preset or full_size or statistics_routes or bf16_forward or large_presets" > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
