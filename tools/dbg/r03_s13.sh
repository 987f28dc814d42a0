#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_lean.py tests/test_gpu_refine.py tests/test_gpu_parity.py tests/test_gpu_dropin_speculation.py tests/test_c_abi.py -q -m gpu -x 2>&1 | tail -8 > $o/s13_tests.log
bash tools/kt_loop.sh 60 > $o/s13_kt_spec.log 2>&1
python tools/call_timeline.py 50 10 2>&1 | head -3 > $o/s13_call50.log
python tools/call_timeline.py 20 10 2>&1 | head -3 > $o/s13_call20.log
