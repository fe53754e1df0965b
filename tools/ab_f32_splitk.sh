# A/B of ND_F32_SPLITK (split-K candidates of the fp32 tuner: conv_mfma_kernel and conv_wino4_kernel) on a bench workload, short
# chains, interleaved on one box; each run tunes for itself (--retune).   bash tools/ab_f32_splitk.sh [config2]
WL=${1:-config2}
for rep in 1 2; do
  for m in 0 1; do
    ND_F32_SPLITK=$m python bench.py --workload $WL --chain 40 --steps 2 --warmup 1 --retune --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
q = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL f32_splitk=$m', q['ms_per_sampler_step'], q['passes']['ms'], 'conv3x3', q['forward']['ms_by_class']['conv3x3'], 'conv1x1', q['forward']['ms_by_class']['conv1x1'])"
  done
done
