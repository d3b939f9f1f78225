#!/bin/bash
# headline step at different chunk counts of the two-stream schedule (0 = automatic); run on the GPU box from the repo root
for c in ${CHUNKS:-0 4 6 8 10}; do
  python bench.py --no-extras --cpu-frames 0 --no-profile --split-chunks $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('chunks', sys.argv[1], round(d['value'] / 1e6, 1), 'M keypoints/s', round(d['ms_per_step'], 3), 'ms')" $c
done
