#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_dropin_speculation.py tests/test_gpu_train_replay.py -q -m gpu -x 2>&1 | tail -5 > $o/s23_tests.log
LOOP_PLAIN=1 bash tools/kt_loop.sh 40 > $o/s23_kt_plain.log 2>&1
python tools/dbg/train_kernels.py 2>&1 | grep -v amdgpu > $o/s23_train.log
python tools/scene_sweep.py 2>&1 | grep -v amdgpu | cut -c1-160 > $o/s23_sweep.log
