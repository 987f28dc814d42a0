#!/bin/bash
# usage (GPU box): tools/dbg/hwq.sh  -- `value` of bench.py (12 frames in flight) against the HIP runtime's GPU_MAX_HW_QUEUES
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for q in default 2 8 12 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for k in 20 50; do
  python bench.py --no-cpu-baseline --no-train-leg --no-cam-leg --repeats 2 --steps $k 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('queues $q K $k value', [round(x) for x in d['value_repeats']], 'single', round(d['single_frame_iters_per_s']))"
  done
done
