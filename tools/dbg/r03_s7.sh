#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
for b in default 2 16; do
  if [ $b != default ]; then export GSR_PBIN_BANDS=$b; fi
  echo "bands $b" >> $o/s7.log
  python tools/dbg/train_kernels.py 2>&1 | grep -v amdgpu >> $o/s7.log
  LOOP_PLAIN=1 python tools/loop_profile.py 2>&1 | tail -3 >> $o/s7.log
done
