#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_refine.py tests/test_gpu_lean.py tests/test_gpu_dropin_speculation.py -q -m gpu -x 2>&1 | tail -60 > $o/s2_tests.log
python -m pytest tests/test_gpu_train_replay.py -q -m gpu -s 2>&1 | grep -E "train replay|passed|failed|Error|assert" | head -20 > $o/s2_train.log
python tools/call_timeline.py 20 10 2>&1 | head -4 > $o/s2_call20.log
python tools/call_timeline.py 50 10 2>&1 | head -4 > $o/s2_call50.log
python tools/localize_split.py --frames 64 > $o/s2_split_near.log 2>&1
python tools/localize_split.py --frames 64 --spread 0.3 10 > $o/s2_split_far.log 2>&1
python bench.py --no-train-leg > $o/s2_bench.log 2>&1
GSR_TIMING=1 python gs_localization_amd/build.py > $o/s2_build_timing.log 2>&1
python tools/phase_timing.py > $o/s2_phase.log 2>&1
