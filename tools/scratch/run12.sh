set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm4 or gemm_1x1 or conv1x1 or stats or fused_groupnorm" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
python tools/time_conv1x1.py "64 32 32 384 384;64 32 32 384 1152;64 16 16 576 576;64 64 64 384 192;64 16 16 576 1728;64 8 8 768 2304" 14,15 > gpurun_out/time_conv1x1_c.log 2>&1
cat gpurun_out/time_conv1x1_c.log
