set -e
mkdir -p gpurun_out
python tools/time_conv1x1.py "64 32 32 384 384;64 32 32 384 1152;64 16 16 576 576;64 64 64 384 192;64 16 16 576 1728" 14,15,5 > gpurun_out/time_conv1x1.log 2>&1
cat gpurun_out/time_conv1x1.log
