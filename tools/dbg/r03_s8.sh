#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o; rm -f $o/s8.log
for b in 1 2 3; do
  export GSR_PBIN_BANDS=$b
  echo "bands $b" >> $o/s8.log
  python tools/scene_sweep.py 2>&1 | grep -v amdgpu | cut -c1-200 >> $o/s8.log
done
