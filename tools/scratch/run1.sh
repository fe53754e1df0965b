set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sampler or copy_row or step" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "hoisted or sampler_loops or one_captured or config4_workload or sharded or trainer" > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 64 > gpurun_out/ops_config2.txt 2>&1
python bench.py --steps 2 --warmup 1 > gpurun_out/b_hoist.json 2> gpurun_out/b_hoist.err
ND_HOIST_EMBED=0 python bench.py --steps 2 --warmup 1 > gpurun_out/b_nohoist.json 2> gpurun_out/b_nohoist.err
python bench.py --steps 2 --warmup 1 > gpurun_out/b_hoist2.json 2> gpurun_out/b_hoist2.err
python - <<'PY'
import json
for f in ('b_hoist','b_nohoist','b_hoist2'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
