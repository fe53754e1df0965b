set -e
mkdir -p gpurun_out
export ND_HIP_LIB=gpurun_variants/libnd_f4DIAG.so
for S in "64 64 64 192 192" "64 64 64 192 192 res" "64 64 64 384 192" "64 32 32 384 384" "64 32 32 768 384"; do
echo "## $S"; python tools/wf4_timeline.py $S 2>&1 | grep -v amdgpu.ids
done > gpurun_out/wf4_timeline.txt
cat gpurun_out/wf4_timeline.txt
