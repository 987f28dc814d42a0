#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_refine.py tests/test_gpu_lean.py tests/test_gpu_dropin_speculation.py -q -m gpu -x 2>&1 | tail -4 > $o/s18_tests.log
python tools/refine_call_timing.py 2>&1 | grep -E "iterations:|fallbacks" | cut -c1-150 > $o/s18_far.log
python tools/localize_split.py --frames 64 --spread 0.3 10 2>&1 | tail -1 | cut -c1-400 > $o/s18_split_far.log
python tools/localize_split.py --frames 64 2>&1 | tail -1 | cut -c1-400 > $o/s18_split_near.log
python tools/call_timeline.py 50 10 2>&1 | head -3 > $o/s18_call50.log
