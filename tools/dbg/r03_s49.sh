#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_refine.py tests/test_gpu_lean.py tests/test_gpu_deterministic.py tests/test_gpu_multirank.py -q -m gpu -x 2>&1 | tail -4 > $o/s49_tests.log
python tools/call_timeline.py 20 10 2>&1 | grep "K =" > $o/s49_calls.log
python tools/call_timeline.py 50 10 2>&1 | grep "K =" >> $o/s49_calls.log
python bench.py --steps 20 --no-cpu-baseline --no-train-leg --repeats 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('K20: value', round(d['value']), 'single', round(d['single_frame_iters_per_s']), 'overhead', round(d['per_call_overhead_ms'], 3))
" >> $o/s49_calls.log
python tools/localize_split.py --frames 64 2>&1 | tail -1 | cut -c1-300 >> $o/s49_calls.log
