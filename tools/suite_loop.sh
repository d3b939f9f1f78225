#!/bin/bash
# N consecutive full `-m gpu` runs ("100 consecutive full-suite runs green"); stops at the first failure and keeps its log.
cd "$(dirname "$0")/.."
N=${1:-100}
OUT=gpurun_out/suite_loop
mkdir -p "$OUT"
: > "$OUT/summary.txt"
for i in $(seq "$N"); do
    AFV_STRESS_ROUNDS=${AFV_STRESS_ROUNDS:-60} timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/run.log" 2>&1
    rc=$?
    echo "run $i rc $rc: $(grep -E 'passed|failed' "$OUT/run.log" | tail -1)" >> "$OUT/summary.txt"
    if [ $rc -ne 0 ]; then cp "$OUT/run.log" "$OUT/failed_run_$i.log"; break; fi
done
echo "sources $(python tools/csrc_sha.py 2>/dev/null)" >> "$OUT/summary.txt"
tail -3 "$OUT/summary.txt"
