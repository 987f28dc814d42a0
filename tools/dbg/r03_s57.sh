#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
for P in 3000000 8000000; do
  for env in "" "GSR_NO_GROUPS=1"; do
    echo "P=$P $env" >> $o/s57_big.log
    env $env GAUSSIANS=$P ITERS=40 timeout 900 python tools/dbg/big_map.py 2>&1 | grep -v amdgpu | cut -c1-200 >> $o/s57_big.log
  done
done
