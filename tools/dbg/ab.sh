#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage: tools/dbg/ab.sh "<defs A>" "<defs B>" ...  -- alternates builds (GSR_DEFS), prints the speculative loop's wall time and kernel times
for v in "$@"; do
  export GSR_LIB_PATH=/tmp/gsr_variant.so; GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v]"
  timeout 120 python tools/loop_profile.py 2>&1 | grep -v amdgpu.ids | grep "spec True" | tail -1 | grep -o "wall ms/iter [0-9.]*\|'preprocess_fwd': [0-9.]*\|'render_fwd': [0-9.]*\|'render_bwd': [0-9.]*\|'preprocess_bwd': [0-9.]*" | paste - - - - -
done
