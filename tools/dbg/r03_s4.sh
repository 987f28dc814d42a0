#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_c_abi.py tests/test_gpu_parity.py tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -15 > $o/s4_tests.log
LOOP_PLAIN=1 bash tools/kt_loop.sh 40 > $o/s4_kt_plain.log 2>&1
bash tools/kt_loop.sh 40 > $o/s4_kt_spec.log 2>&1
python tools/train_step_bench.py > $o/s4_train.log 2>&1
