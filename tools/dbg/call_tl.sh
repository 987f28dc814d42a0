#!/bin/bash
# usage (GPU box): tools/dbg/call_tl.sh [K]  -- one refine() call on the GPU's timeline (kernel trace taken apart by tools/kt_calls.py)
# and the host-side split of the per-call overhead (tools/call_timeline.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=${1:-20}; o=gpurun_out/r04; mkdir -p $o
python tools/call_timeline.py $K 10 2>/dev/null | head -30 > $o/call_host_$K.log
rocprofv3 --kernel-trace -d $o/ctl -o ctl --output-format csv -- python3 tools/call_timeline.py $K 6 > /dev/null 2>&1
python tools/kt_calls.py "$(find $o/ctl -name ctl_kernel_trace.csv | head -1)" > $o/call_gpu_$K.log 2>&1
head -12 $o/call_host_$K.log; head -60 $o/call_gpu_$K.log
