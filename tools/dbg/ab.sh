#!/bin/bash
# usage: tools/dbg/ab.sh "<defs A>" "<defs B>" ...  -- alternates builds, prints loop wall times and binning kernel times
for v in "$@"; do
  GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v]"
  timeout 120 python tools/loop_profile.py 2>&1 | grep -v amdgpu.ids | head -1 | grep -o "spec [A-Za-z]* wall ms/iter [0-9.]*\|'tile_count[^}]*'render_fwd': [0-9.]*" | paste - -
  timeout 100 python tools/dbg/train_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "'tile_count[^}]*'render_fwd': [0-9.]*"
done
