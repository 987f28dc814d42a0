#!/bin/bash
# usage: tools/dbg/ab.sh "<defs A>" "<defs B>" ...  -- alternates builds (GSR_DEFS), prints the speculative loop's wall time and kernel times
for v in "$@"; do
  GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v]"
  timeout 120 python tools/loop_profile.py 2>&1 | grep -v amdgpu.ids | grep "spec True" | tail -1 | grep -o "wall ms/iter [0-9.]*\|'preprocess_fwd': [0-9.]*\|'render_fwd': [0-9.]*\|'render_bwd': [0-9.]*\|'preprocess_bwd': [0-9.]*" | paste - - - - -
done
