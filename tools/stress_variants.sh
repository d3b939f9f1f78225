#!/bin/bash
# round-6 hunt for the one-in-hundreds wrong descriptor of the three-thread scene: the stress test against variant builds of the library
#   python tools/experiments.py build nozc=AFV_NO_ZERO_COPY kp1=AFV_KP_PER_BLOCK=1 strong=AFV_DESCRIBE_STRONG_SYNC   (here)
#   bash tools/stress_variants.sh base nozc kp1 strong                                                              (on the box)
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-300}
REPS=${REPS:-3}
for v in "$@"; do
    if [ "$v" = base ]; then LIB=""; else LIB=$PWD/anyfeature-vslam_amd/build_exp/libafv_$v.so; fi
    for k in $(seq $REPS); do
        out=$(AFV_TEST_LIB=$LIB AFV_STRESS_ROUNDS=$ROUNDS timeout 300 python -m pytest tests/test_gpu_match.py -m gpu -q -s -p no:cacheprovider -k three_threads 2>&1)
        echo "$v run $k: $(echo "$out" | tail -1) $(echo "$out" | grep '^three-threads' | cut -c1-400 | head -2)"
    done
done
