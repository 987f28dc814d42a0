#!/bin/bash
# usage (GPU box): tools/dbg/flight.sh  -- `value` of bench.py against frames in flight (one HIP hardware queue each, at most 16)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 8 12 16 20 24; do
  for k in 20 50; do
  python bench.py --no-cpu-baseline --no-train-leg --no-cam-leg --repeats 2 --steps $k --frames-in-flight $f 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('frames $f K $k queues', d['config']['hip_hw_queues'], 'value', [round(x) for x in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']))"
  done
done
