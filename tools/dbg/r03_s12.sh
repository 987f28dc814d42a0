#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_lean.py tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -8 > $o/s12_tests.log
bash tools/kt_loop.sh 60 > $o/s12_kt_spec.log 2>&1
python tools/call_timeline.py 50 10 2>&1 | head -3 > $o/s12_call50.log
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
GSR_TIM_DUMP=/tmp/tim_rows.txt python tools/phase_timing.py > $o/s12_phase.log 2>&1
python tools/dbg/lean_tail.py /tmp/tim_rows.txt 40 > $o/s12_lean_tail.log 2>&1
