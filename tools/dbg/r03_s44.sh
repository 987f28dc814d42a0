#!/bin/bash
# k_preprocess_lean at 4 / 5 / 6 waves per SIMD (register cap): does a second frame's kernel fit next to it?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
for occ in 4 5 6 4 5; do
  GSR_DEFS="-DGSR_LEAN_OCC=$occ" python gs_localization_amd/build.py > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-train-leg --repeats 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('occ $occ', 'value', round(d['value']), [round(v) for v in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']), 'lean us', d['kernels_ms_per_iter_native_single_frame']['preprocess_fwd'])
" >> $o/s44_occ.log
done
