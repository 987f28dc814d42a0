#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu 2>&1 | tail -5 > $o/s59_suite.log
python bench.py --no-cpu-baseline --no-train-leg --repeats 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value']), [round(v) for v in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']), 'plain', round(d['plain_loop_iters_per_s']), d['kernels_ms_per_iter_native_single_frame'])
" > $o/s59_bench.log
CASES=200 SEED=61 timeout 900 python tools/fuzz_speculation.py 2>&1 | tail -1 > $o/s59_fuzz.log
