#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $o/s3_tests.log
python -m pytest tests/test_gpu_train_replay.py -q -m gpu -s 2>&1 | grep -E "train replay|passed|failed" | head -20 > $o/s3_train.log
python bench.py > $o/s3_bench.log 2>&1
python tools/call_timeline.py 20 10 2>&1 | head -4 > $o/s3_call20.log
rm -rf $o/kt_calls; timeout 600 rocprofv3 --kernel-trace -d $o/kt_calls -o k --output-format csv -- python3 tools/call_timeline.py 20 6 > $o/s3_kt.log 2>&1
python3 tools/kt_calls.py "$(find $o/kt_calls -name 'k_kernel_trace.csv' | head -1)" 20 > $o/s3_kt_calls.log 2>&1
rm -rf $o/kt_calls
LOOP_PLAIN=1 bash tools/kt_loop.sh 40 > $o/s3_kt_plain.log 2>&1
