#!/bin/bash
# round 6: the three concurrency tests on the current sources and on the round-5 library (anyfeature-vslam_amd/build_exp/libafv_r05.so, built
# from commit 10eec22): they must pass on the first and fail on the second.
cd "$(dirname "$0")/.."
K="extraction_beside or fresh_contexts or three_threads"
echo "== current sources"
python -m pytest tests/test_gpu_match.py -m gpu -q -p no:cacheprovider -k "$K" 2>&1 | tail -2
if [ -f anyfeature-vslam_amd/build_exp/libafv_r05.so ]; then
    echo "== round-5 library"
    AFV_TEST_LIB=$PWD/anyfeature-vslam_amd/build_exp/libafv_r05.so python -m pytest tests/test_gpu_match.py -m gpu -q -p no:cacheprovider -k "$K" 2>&1 |
        grep -E "differs|Error|passed|failed|error" | cut -c1-260 | head -12
fi
