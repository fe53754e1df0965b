set -e
mkdir -p gpurun_out
export ND_HIP_LIB=gpurun_variants/libnd_f4DIAG.so
for S in "64 64 64 192 192" "64 32 32 384 384"; do
echo "## $S"; python tools/wf4_timeline.py $S 2>&1 | grep -v amdgpu.ids
done
