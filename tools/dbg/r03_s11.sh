#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
GSR_TIM_DUMP=/tmp/tim_rows.txt python tools/phase_timing.py > $o/s11_phase.log 2>&1
python tools/dbg/lean_tail.py /tmp/tim_rows.txt 40 > $o/s11_lean_tail.log 2>&1
