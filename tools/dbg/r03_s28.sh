#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_deterministic.py -q -m gpu -s 2>&1 | tail -40 > $o/s28_det.log
python -m pytest tests/test_gpu_refine.py tests/test_gpu_lean.py tests/test_c_abi.py -q -m gpu -x 2>&1 | tail -5 > $o/s28_tests.log
