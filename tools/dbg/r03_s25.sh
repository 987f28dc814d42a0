#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 > $o/s25_tests.log
LOOP_PLAIN=1 bash tools/kt_loop.sh 40 2>&1 | head -4 > $o/s25_kt_plain.log
python tools/dbg/train_kernels.py 2>&1 | grep -v amdgpu > $o/s25_train.log
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
LOOP_PLAIN=1 python tools/phase_timing.py 2>&1 | sed -n 3,16p > $o/s25_phase_plain.log
