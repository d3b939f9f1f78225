#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/crash
ulimit -c 0
for k in $(seq ${REPS:-8}); do
    AFV_STRESS_ROUNDS=${ROUNDS:-300} timeout 300 python -X faulthandler -m pytest tests/test_gpu_match.py -m gpu -q -s -p no:cacheprovider -k three_threads > gpurun_out/crash/run$k.log 2>&1
    rc=$?
    echo "run $k rc $rc: $(grep -c . gpurun_out/crash/run$k.log) lines; $(grep -i 'fault\|abort\|HSA_STATUS\|Segmentation\|core' gpurun_out/crash/run$k.log | head -3 | cut -c1-300)"
done
