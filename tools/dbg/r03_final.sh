#!/bin/bash
# round-3 closing measurements: full GPU suite, bench (default + K=20), split driver, call timelines under a kernel trace,
# phase clocks (timing build, last: it replaces the .so).  The rocprofv3 counter passes are tools/profile_round.sh.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03/final; mkdir -p $o
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $o/tests.log
python bench.py > $o/bench.log 2>&1
python bench.py --steps 20 --no-cpu-baseline --no-train-leg > $o/bench20.log 2>&1
python tools/localize_split.py --frames 64 > $o/split_near.log 2>&1
python tools/localize_split.py --frames 64 --spread 0.3 10 > $o/split_far.log 2>&1
python tools/scene_sweep.py 2>&1 | grep -v amdgpu | cut -c1-220 > $o/scene_sweep.log
for K in 20 50; do
  rm -rf $o/kt; timeout 600 rocprofv3 --kernel-trace -d $o/kt -o k --output-format csv -- python3 tools/call_timeline.py $K 6 > $o/call${K}_host.log 2>&1
  python3 tools/kt_calls.py "$(find $o/kt -name 'k_kernel_trace.csv' | head -1)" $K > $o/call${K}_kernels.log 2>&1
done
rm -rf $o/kt
python tools/call_timeline.py 20 10 2>&1 | head -3 > $o/call20.log
python tools/call_timeline.py 50 10 2>&1 | head -3 > $o/call50.log
GSR_TIMING=1 python gs_localization_amd/build.py > /dev/null 2>&1
GSR_TIM_DUMP=/tmp/tim_rows.txt python tools/phase_timing.py > $o/phase_spec.log 2>&1
python tools/dbg/lean_tail.py /tmp/tim_rows.txt 40 > $o/lean_tail.log 2>&1
LOOP_PLAIN=1 python tools/phase_timing.py > $o/phase_plain.log 2>&1
