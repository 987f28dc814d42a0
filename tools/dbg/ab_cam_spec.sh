#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box): tools/dbg/ab_cam_spec.sh "<defs A>" "<defs B>" ...  -- same-box A/B of builds (GSR_DEFS) on the speculative loop of
# S-3M-cam (852x480 and 1024x576) and S-1M-640: it/s of a 200-iteration call (tools/loop_only.py); each variant twice, interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  export GSR_LIB_PATH=/tmp/gsr_variant.so; GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo -n "variant [$v] rep $rep: "
  for sc in s_3m_cam s_3m_cam_1024 s_1m_640; do SCENE=$sc python tools/loop_only.py 200 2>/dev/null | tail -1 | cut -d' ' -f1,4-6 | tr '\n' ' '; done; echo
done
done
