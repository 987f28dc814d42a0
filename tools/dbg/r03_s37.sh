#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests/test_gpu_deterministic.py -q -m gpu 2>&1 | tail -15 > $o/s37_det.log
CASES=400 SEED=11 timeout 1500 python tools/fuzz_speculation.py > $o/s37_fuzz11.log 2>&1
CASES=400 SEED=12 timeout 1500 python tools/fuzz_speculation.py > $o/s37_fuzz12.log 2>&1
