set -e
mkdir -p gpurun_out
for i in 1 2; do
for v in f4NEW f4PREV; do
ND_HIP_LIB=gpurun_variants/libnd_$v.so python bench.py --steps 2 --warmup 1 > gpurun_out/b_$v$i.json 2> gpurun_out/b_$v$i.err
done; done
python - <<'PY'
import json
for f in ('b_f4NEW1','b_f4PREV1','b_f4NEW2','b_f4PREV2'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
