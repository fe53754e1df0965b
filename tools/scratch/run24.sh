set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "winograd_f4" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
for S in "64 64 64 192 192" "64 32 32 384 384"; do
echo "## $S"; ND_HIP_LIB=gpurun_variants/libnd_f4DIAG.so python tools/wf4_timeline.py $S 2>&1 | grep "prologue in parts\|per round"
done
S="64 64 64 192 192;64 64 64 384 192;64 64 64 384 384;64 32 32 384 384;64 32 32 768 384;64 16 16 576 576;64 8 8 768 768"
python tools/ab_wf4.py gpurun_variants/libnd_f4WARM.so,gpurun_variants/libnd_f4COLD.so "$S" 5 stats 2>&1 | grep -v amdgpu.ids
python tools/ab_wf4.py gpurun_variants/libnd_f4WARM.so,gpurun_variants/libnd_f4COLD.so "$S" 5 stats,res 2>&1 | grep -v amdgpu.ids
