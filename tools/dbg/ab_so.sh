#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage: tools/dbg/ab_so.sh build_ab/base.so build_ab/pk.so ...  -- same-box A/B of prebuilt libraries: bench value / single-frame
# loop and the native loop's kernel times, alternating twice
for rep in 1 2; do
for v in "$@"; do
  export GSR_LIB_PATH="$v"
  echo "variant [$v] rep $rep"
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg --repeats 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('value', round(d['value']), 'repeats', [round(x) for x in d.get('value_repeats', [])], 'single', round(d.get('single_frame_iters_per_s', 0)), 'plain', round(d.get('plain_loop_iters_per_s', 0)))
"
  timeout 120 python tools/loop_profile.py 2>&1 | grep -v amdgpu.ids | grep "spec True" | head -1 | grep -o "wall ms/iter [0-9.]*\|'render_fwd': [0-9.]*\|'render_bwd': [0-9.]*" | paste - - -
done
done
