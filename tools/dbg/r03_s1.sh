#!/bin/bash
# round-3 GPU session 1: tolerance-audit tests, call timeline + kernel trace, K8 phase clocks (timing build last: it replaces the .so)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_replay.py tests/test_gpu_lean.py -q -m gpu -s 2>&1 | tail -40 > $o/s1_tests.log
python tools/call_timeline.py 20 10 > $o/s1_call20.log 2>&1
python tools/call_timeline.py 50 10 > $o/s1_call50.log 2>&1
rm -rf $o/kt_calls; timeout 600 rocprofv3 --kernel-trace -d $o/kt_calls -o k --output-format csv -- python3 tools/call_timeline.py 20 6 > $o/s1_kt.log 2>&1
python3 tools/kt_calls.py "$(find $o/kt_calls -name 'k_kernel_trace.csv' | head -1)" 20 > $o/s1_kt_calls.log 2>&1
rm -rf $o/kt_calls
GSR_TIMING=1 python gs_localization_amd/build.py > $o/s1_build_timing.log 2>&1
python tools/phase_timing.py > $o/s1_phase.log 2>&1
