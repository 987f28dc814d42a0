#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
for i in 1 2 3; do python -m pytest "tests/test_gpu_lean.py::test_native_loop_at_baseline_size" -q -m gpu 2>&1 | grep -E "AssertionError|assert |passed|failed|^E " | head -12; done > $o/s29.log
