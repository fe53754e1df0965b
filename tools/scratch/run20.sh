set -e
mkdir -p gpurun_out
S="64 64 64 192 192;64 64 64 384 192;64 32 32 384 384;64 32 32 768 384"
L=gpurun_variants/libnd_f4BASE.so,gpurun_variants/libnd_f4NOPW.so,gpurun_variants/libnd_f4NOEPI.so,gpurun_variants/libnd_f4NOEPW.so
ND_AB_NOCHECK=1 python tools/ab_wf4.py $L "$S" 5 stats > gpurun_out/wf4_abl_20.log 2>&1
cat gpurun_out/wf4_abl_20.log
