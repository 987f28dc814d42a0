#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $o/s48_suite.log
GSR_DETERMINISTIC=1 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $o/s48_suite_det.log
CASES=200 SEED=41 timeout 1200 python tools/fuzz_speculation.py 2>&1 | tail -2 > $o/s48_fuzz.log
GSR_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-train-leg --repeats 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('det: value', round(d['value']), 'single', round(d['single_frame_iters_per_s']), 'plain', round(d['plain_loop_iters_per_s']), d['kernels_ms_per_iter_native_single_frame'])
" > $o/s48_bench_det.log
