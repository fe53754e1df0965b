mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_full.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_tests_full.log
exit $rc
