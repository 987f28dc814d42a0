#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box): tools/dbg/ab_plain.sh "<defs A>" "<defs B>" ...  -- same-box A/B of builds (GSR_DEFS) on the complete-list path:
# plain-loop it/s on S-1M-640 / S-3M-cam and the train step's kernel times; each variant twice, interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  export GSR_LIB_PATH=/tmp/gsr_variant.so; GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v] rep $rep"
  for sc in s_1m_640 s_3m_cam; do SCENE=$sc LOOP_PLAIN=1 python tools/loop_only.py 150 2>/dev/null | tail -1 | cut -c1-60; done
done
done
