#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_lean.py tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -12 > $o/s54_tests.log
for env in "" "GSR_NO_GROUPS=1"; do
env $env python bench.py --no-cpu-baseline --no-train-leg --repeats 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$env value', round(d['value']), [round(v) for v in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']), 'plain', round(d['plain_loop_iters_per_s']), d['kernels_ms_per_iter_native_single_frame'])
" >> $o/s54_bench.log; done
python tools/scene_sweep.py 2>&1 | grep -v amdgpu | cut -c1-200 > $o/s54_sweep.log
CASES=150 SEED=51 timeout 900 python tools/fuzz_speculation.py 2>&1 | tail -2 > $o/s54_fuzz.log
