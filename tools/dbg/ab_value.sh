#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box): tools/dbg/ab_value.sh K "<defs A>" "<defs B>" ...  -- same-box A/B of builds (GSR_DEFS) on bench.py's `value`
# (K iterations per call), each variant twice, interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=$1; shift
for rep in 1 2; do
for v in "$@"; do
  export GSR_LIB_PATH=/tmp/gsr_variant.so; GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo -n "variant [$v] rep $rep: "
  python bench.py --gpus 1 --steps $K --warmup 5 --no-cpu-baseline --no-train-leg --no-cam-leg 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1
done
done
