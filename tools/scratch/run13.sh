set -e
mkdir -p gpurun_out
L=nice-diffusion_amd/nicediffusion/libnd_hip.so
: > gpurun_out/wf4_ngroup.log
for g in 0 1 2 3 4; do
  echo "## ND_NGROUP=$g (0 = the library's choice)" >> gpurun_out/wf4_ngroup.log
  ND_NGROUP=$g ND_AB_SPLITS=2 python tools/ab_wf4.py $L "64 16 16 576 576;64 8 8 768 768;64 16 16 1152 576;64 8 8 1536 768" 5 stats >> gpurun_out/wf4_ngroup.log 2>&1
  ND_NGROUP=$g ND_AB_SPLITS=1 python tools/ab_wf4.py $L "64 16 16 576 576;64 16 16 1152 576;64 32 32 384 384;64 32 32 768 384;64 64 64 192 192;64 64 64 384 192" 5 stats >> gpurun_out/wf4_ngroup.log 2>&1
done
grep -v amdgpu.ids gpurun_out/wf4_ngroup.log
