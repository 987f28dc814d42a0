#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
LOOP_PLAIN=1 python tools/phase_timing.py 2>&1 | sed -n 2,18p > $o/s24_phase_plain.log
