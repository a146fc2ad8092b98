#!/bin/bash
# Runs on the GPU box: parity tests of the batch tracers with the work-queue schedule included (shipped build) and with a
# merged-rounds build of the streaming tracer, then both scenes in HBM with the shipped build and the merged builds side by side.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
RT_TEST_QUEUE=1 timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_trace.py tests/test_gpu_configs.py -m gpu > gpurun_out/r03_queue_tests.log 2>&1 || { echo TESTS queue FAILED; tail -30 gpurun_out/r03_queue_tests.log; exit 1; }
tail -3 gpurun_out/r03_queue_tests.log
for V in merged1 merged12; do
RT_AMD_DEVICE_LIB=$R/build_exp/lib_$V.so timeout -k 10 500 python3 -u -m pytest -x -q --timeout 200 --timeout-method thread tests/test_gpu_trace.py tests/test_gpu_configs.py -m gpu > gpurun_out/r03_${V}_tests.log 2>&1 || { echo TESTS $V FAILED; tail -20 gpurun_out/r03_${V}_tests.log; exit 1; }
tail -3 gpurun_out/r03_${V}_tests.log
done
tools/variants_run.sh soup primary,incoherent,path default merged1 merged6 merged12 && tools/variants_run.sh grid primary,incoherent,path default merged1 merged12
