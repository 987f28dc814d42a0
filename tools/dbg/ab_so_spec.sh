#!/bin/bash
# usage (GPU box): tools/dbg/ab_so_spec.sh a.so b.so ...  -- same-box A/B of prebuilt libraries on the speculative loop (tools/loop_profile.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  cp "$v" gs_localization_amd/libgsr_hip.so; touch gs_localization_amd/libgsr_hip.so
  echo -n "[$v] rep $rep  "
  timeout 300 python tools/loop_profile.py 2>&1 | grep -v amdgpu.ids | grep "spec True" | tail -1 | grep -o "wall ms/iter [0-9.]*\|'preprocess_fwd': [0-9.]*\|'render_fwd': [0-9.]*\|'render_bwd': [0-9.]*\|'preprocess_bwd': [0-9.]*" | paste - - - - -
done
done
