#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $o/s30_tests.log
python bench.py --no-cpu-baseline --no-train-leg > $o/s30_bench.log 2>&1
GSR_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-train-leg > $o/s30_bench_det.log 2>&1
