mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -s > gpurun_out/r05_gpu_tests_printed_numbers.txt 2>&1
rc=$?
tail -3 gpurun_out/r05_gpu_tests_printed_numbers.txt
[ $rc = 0 ] || exit $rc
PMC_WORKLOADS="" ROUND=r05 bash tools/profile_round.sh > gpurun_out/profile_round_a.log 2>&1
tail -12 gpurun_out/profile_round_a.log
