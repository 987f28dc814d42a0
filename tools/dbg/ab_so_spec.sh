#!/bin/bash
# (variant builds go to GSR_LIB_PATH and are loaded from there: the product library is never overwritten -- build.py, _lib.py)
# usage (GPU box): tools/dbg/ab_so_spec.sh a.so b.so ...  -- same-box A/B of prebuilt libraries on the speculative loop (tools/loop_profile.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  export GSR_LIB_PATH="$v"
  echo -n "[$v] rep $rep  "
  timeout 300 python tools/loop_profile.py 2>&1 | grep -v amdgpu.ids | grep "spec True" | tail -1 | grep -o "wall ms/iter [0-9.]*\|'preprocess_fwd': [0-9.]*\|'render_fwd': [0-9.]*\|'render_bwd': [0-9.]*\|'preprocess_bwd': [0-9.]*" | paste - - - - -
done
done
