#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box): tools/dbg/ab_train.sh "<defs A>" "<defs B>" ...  -- same-box A/B of builds (GSR_DEFS) on the >2048-tile path:
# per-kernel HIP-event times of the train step (1296x840; TK_CASES="WxH:P ..." TK_LINES=n: other sizes; TK_STEP=1: also the wall time of a whole train step) and the plain loop at 1024x576
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export GSR_LIB_PATH=/tmp/gsr_variant.so; GSR_DEFS="$v" python gs_localization_amd/build.py > /dev/null 2>&1
  echo "variant [$v]"
  python tools/dbg/train_kernels.py $TK_CASES 2>/dev/null | tail -${TK_LINES:-1} | cut -c1-270
  [ -n "$TK_STEP" ] && python tools/train_only.py 30 2>/dev/null | tail -1
  [ -z "$TK_CASES" ] && SCENE=s_3m_cam_1024 LOOP_PLAIN=1 python tools/loop_only.py 100 2>/dev/null | tail -1 | cut -c1-50
done
