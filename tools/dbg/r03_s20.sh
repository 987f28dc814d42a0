#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $o/s20_tests.log
python tools/python_loop_cprofile.py 2>&1 | head -16 > $o/s20_pyprof.log
python bench.py --no-cpu-baseline --no-train-leg 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value']), 'single', round(d['single_frame_iters_per_s']), 'plain', round(d['plain_loop_iters_per_s']), 'python', round(d['python_loop_iters_per_s']))
" > $o/s20_bench.log
