#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o; rm -f $o/s14.log
for F in 8 12 16; do
  python bench.py --frames-in-flight $F --no-cpu-baseline --no-train-leg 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('F', d['config']['frames_in_flight_per_gpu'], 'value', round(d['value']), [round(x) for x in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']), 'plain', round(d['plain_loop_iters_per_s']), 'python', round(d['python_loop_iters_per_s']))
" >> $o/s14.log
done
