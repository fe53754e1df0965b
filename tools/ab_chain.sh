#!/bin/bash
# tools/ab_chain.sh WORKLOAD CHAIN ROUNDS LIB_A LIB_B [...]: interleaved same-box A/B of library BUILDS on short chains of a bench
# workload (hipGraph replay, the committed tune cache -- a variant build takes the product build's): one bench.py process per
# (round, lib), ms per sampler step of each, medians at the end.  LIB = path of a .so, or "base" = the library in the tree.
set -u
WL=$1; CH=$2; RO=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
export ND_ALLOW_ABLATION=1 ND_TUNE_STAMP_ANY=1
O=$R/gpurun_out/ab_chain_$$.log; : > $O
for r in $(seq 1 $RO); do
  for L in "$@"; do
    if [ "$L" = base ]; then unset ND_HIP_LIB; else export ND_HIP_LIB=$R/$L; fi
    v=$(timeout -k 10 300 python3 $R/bench.py --workload $WL --chain $CH --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(l['ms_per_sampler_step'], l['config']['tune_cache']['choices_loaded'])")
    echo "$r $L $v" | tee -a $O
  done
done
python3 - $O <<'PY'
import sys, statistics, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) >= 3: d[p[1]].append(float(p[2]))
for k, v in d.items(): print('%-40s median %.3f ms per sampler step  (%s)' % (k, statistics.median(v), ' '.join('%.3f' % x for x in v)))
PY
