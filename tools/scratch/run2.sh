set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "first or last_conv or gather" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "tiny_forward or preset or full_size or sampler_loops or hoisted" > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
ND_LAYER_TABLE_OPS=1 python tools/layer_table.py 64 > gpurun_out/ops_config2_edge.txt 2>&1
python bench.py --steps 2 --warmup 1 > gpurun_out/b_edge.json 2> gpurun_out/b_edge.err
ND_EDGE_CONVS=0 python bench.py --steps 2 --warmup 1 > gpurun_out/b_noedge.json 2> gpurun_out/b_noedge.err
python bench.py --steps 2 --warmup 1 > gpurun_out/b_edge2.json 2> gpurun_out/b_edge2.err
python - <<'PY'
import json
for f in ('b_edge','b_noedge','b_edge2'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
