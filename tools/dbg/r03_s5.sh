#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
LOOP_PLAIN=1 bash tools/kt_loop.sh 40 > $o/s5_kt_plain.log 2>&1
python tools/train_step_bench.py 2>&1 | tail -5 > $o/s5_train.log
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
LOOP_PLAIN=1 python tools/phase_timing.py > $o/s5_phase_plain.log 2>&1
