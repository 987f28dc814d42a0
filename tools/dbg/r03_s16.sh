#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o; rm -f $o/s16.log
for L in 1 2 4; do
  GSR_DEFS="-DGSR_LEAN_POOL=$L" python gs_localization_amd/build.py > /dev/null 2>&1
  python -m pytest tests/test_gpu_lean.py -q -m gpu -x 2>&1 | tail -1 >> $o/s16.log
  python bench.py --no-cpu-baseline --no-train-leg 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pool $L: value', round(d['value']), [round(x) for x in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']), 'lean us', d['kernels_ms_per_iter_native_single_frame']['preprocess_fwd'])
" >> $o/s16.log
done
