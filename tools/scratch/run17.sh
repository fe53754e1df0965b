set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "groupnorm" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
python tools/layer_table.py 64 2>&1 | tail -1 > gpurun_out/gn_coef_on.txt
ND_GN_APPLY_COEFFS=0 python tools/layer_table.py 64 2>&1 | tail -1 > gpurun_out/gn_coef_off.txt
python tools/layer_table.py 64 2>&1 | tail -1 >> gpurun_out/gn_coef_on.txt
cat gpurun_out/gn_coef_on.txt gpurun_out/gn_coef_off.txt
python bench.py --steps 2 --warmup 1 > gpurun_out/b_coef_on.json 2> gpurun_out/b_coef_on.err
ND_GN_APPLY_COEFFS=0 python bench.py --steps 2 --warmup 1 > gpurun_out/b_coef_off.json 2> gpurun_out/b_coef_off.err
python bench.py --steps 2 --warmup 1 > gpurun_out/b_coef_on2.json 2> gpurun_out/b_coef_on2.err
python - <<'PY'
import json
for f in ('b_coef_on','b_coef_off','b_coef_on2'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py -x -q -m gpu -k "tiny_forward or preset or full_size or statistics_routes or bf16_forward or large_presets or sampler_loops" > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
