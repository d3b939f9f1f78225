#!/bin/bash
# Differential run of the whole `-m gpu` suite on the TEST-ONLY poison build (csrc/afv_poison.h, tools/poison_build.py): every
# allocation and the LDS of every CU filled with 0xA5, then with 0x5A.  A test that fails in either pass, or whose outcome differs
# between the passes, names a kernel that reads memory it did not write.  On the GPU box:   bash tools/poison_suite.sh [rounds]
# Writes gpurun_out/poison/{a5,5a,stress_*}.log and a one-line verdict per pass.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/poison
mkdir -p "$OUT"
LIB=$PWD/anyfeature-vslam_amd/build_exp/libafv_poison.so
if [ ! -f "$LIB" ] || [ "$(cat "${LIB%.so}.sha" 2>/dev/null)" != "$(python tools/csrc_sha.py)" ]; then  # absent, or built from other sources
    python tools/poison_build.py > "$OUT/build.log" 2>&1 || { echo "poison build failed"; exit 1; }
fi
verdict() { grep -E "passed|failed|error" "$1" | tail -1; }
ROUNDS=${1:-100}
for b in 0xA5 0x5A; do
    AFV_TEST_LIB=$LIB AFV_POISON_BYTE=$b AFV_POISON_VERBOSE=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/suite_$b.log" 2>&1
    echo "poison $b: rc $? : $(verdict "$OUT/suite_$b.log")"
done
# the three-thread scene, many rounds, INSIDE a process that ran the other matcher tests first (stale pages, warm allocator)
AFV_STRESS_ROUNDS=$ROUNDS timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_extract.py -m gpu -q -p no:cacheprovider > "$OUT/stress_plain.log" 2>&1
echo "stress plain x$ROUNDS: rc $? : $(verdict "$OUT/stress_plain.log")"
for b in 0xA5 0x5A; do
    AFV_TEST_LIB=$LIB AFV_POISON_BYTE=$b AFV_STRESS_ROUNDS=$ROUNDS timeout 900 python -m pytest tests/test_gpu_match.py -m gpu -q -p no:cacheprovider > "$OUT/stress_$b.log" 2>&1
    echo "stress poison $b x$ROUNDS: rc $? : $(verdict "$OUT/stress_$b.log")"
done
