set -e
mkdir -p gpurun_out
python bench.py --workload config2 --steps 2 --warmup 1 --retune --save-tune-cache gpurun_out/tune_cache_config2.json > gpurun_out/b_retune.json 2> gpurun_out/b_retune.err
cp gpurun_out/tune_cache_config2.json profiles/tune_cache_config2.json
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 64 > gpurun_out/ops_config2_s1.txt 2>&1
python bench.py --steps 3 --warmup 1 > gpurun_out/b_s1.json 2> gpurun_out/b_s1.err
python - <<'PY'
import json
for f in ('b_retune','b_s1'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
grep -c . gpurun_out/ops_config2_s1.txt
grep "channel_partials\|conv1x1_stats" gpurun_out/ops_config2_s1.txt
