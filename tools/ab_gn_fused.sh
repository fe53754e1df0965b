# A/B of ND_GN_FUSED_MAX (the one-launch GroupNorm threshold): short chains of a bench workload, alternating, same box.
#   bash tools/ab_gn_fused.sh config2 "1048576 8388608"
WL=${1:-config2}
TH=${2:-"1048576 4194304 8388608"}
CH=40; [ $WL = config5 ] && CH=50
for rep in 1 2; do
for m in $TH; do
    ND_GN_FUSED_MAX=$m python bench.py --workload $WL --chain $CH --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "
import sys, json
q = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL fused_max=$m', q['ms_per_sampler_step'], q['passes']['ms'])"
done
done
