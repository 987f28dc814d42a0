#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -3 > $o/s45_tests.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-train-leg --repeats 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value']), [round(v) for v in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']), 'plain', round(d['plain_loop_iters_per_s']), d['kernels_ms_per_iter_native_single_frame'])
" >> $o/s45_bench.log; done
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
python tools/phase_timing.py 2>&1 | grep -A 10 "k_preprocess_bwd (cycles" > $o/s45_phase.log
