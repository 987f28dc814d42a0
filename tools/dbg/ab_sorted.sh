#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box): tools/dbg/ab_sorted.sh "<defs A>" "<defs B>" ...  -- same-box A/B of builds (GSR_DEFS) on a map in its random draw
# order and in 3D Morton order (tools/dbg/sort_probe.py): speculative and plain loop of S-1M-640 and S-3M-cam
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export GSR_LIB_PATH=/tmp/gsr_variant.so; GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v]"
  for sc in s_1m_640 s_3m_cam; do for s in 0 1; do SCENE=$sc SORT=$s python tools/dbg/sort_probe.py 2>/dev/null | grep "SORT" | grep "${ONLY:-S}" | cut -c1-200; done; done
done
