# Where do the waves of the dominant Winograd kernel wait?  (separate rocprofv3 --pmc pass; run on the GPU box)
cd /tmp && export TMPDIR=/tmp && export WINO=1
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $R/gpurun_out/pmcwait -o runc --output-format csv -- python3 $R/tools/conv_bench.py 64 32 32 384 384 3 8 3 > $R/gpurun_out/pmcwait.log 2>&1
