#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu 2>&1 | tail -40 > $o/s9_tests.log
python bench.py > $o/s9_bench.log 2>&1
python bench.py --steps 20 --no-cpu-baseline --no-train-leg > $o/s9_bench20.log 2>&1
python tools/localize_split.py --frames 64 > $o/s9_split_near.log 2>&1
python tools/localize_split.py --frames 64 --spread 0.3 10 > $o/s9_split_far.log 2>&1
