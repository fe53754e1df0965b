set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -s > gpurun_out/t_bf16.log 2>&1 || { tail -40 gpurun_out/t_bf16.log; exit 1; }
tail -3 gpurun_out/t_bf16.log
grep "preset OPENAI\|full-batch" gpurun_out/t_bf16.log
for WL in config4 config5; do
python bench.py --workload $WL --steps 2 --warmup 1 > gpurun_out/b_${WL}_taps.json 2> gpurun_out/b_${WL}_taps.err
ND_EDGE_CONVS=0 python bench.py --workload $WL --steps 2 --warmup 1 > gpurun_out/b_${WL}_notaps.json 2> gpurun_out/b_${WL}_notaps.err
done
python - <<'PY'
import json
for f in ('b_config4_taps','b_config4_notaps','b_config5_taps','b_config5_notaps'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_sampler_step'], d['forward']['launches'])
PY
