#!/bin/bash
# value (aggregate iterations/s) against the number of frames in flight on one GPU
for F in 1 2 4 6 8 12; do
  echo -n "frames in flight $F: "
  timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-train-leg --repeats 3 --frames-in-flight $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value']), 'repeats', [round(x) for x in d.get('value_repeats', [])], 'single', round(d.get('single_frame_iters_per_s', 0)))
"
done
