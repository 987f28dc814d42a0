#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $o/s10_tests.log
bash tools/kt_loop.sh 60 > $o/s10_kt_spec.log 2>&1
python tools/call_timeline.py 20 10 2>&1 | head -3 > $o/s10_call20.log
python tools/call_timeline.py 50 10 2>&1 | head -3 > $o/s10_call50.log
python tools/python_loop_cprofile.py 2>&1 | head -45 > $o/s10_pyprof.log
