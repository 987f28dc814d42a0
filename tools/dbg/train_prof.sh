#!/bin/bash
# usage (GPU box): tools/dbg/train_prof.sh [tag]  -- rocprofv3 kernel stats of 20 train steps (tools/train_only.py), this library's kernels only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=${1:-x}; o=gpurun_out/r04/tp_$t; mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o -o tp --output-format csv -- python3 tools/train_only.py 20 > $o/run.log 2>&1
tail -1 $o/run.log
python - "$o" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/tp_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "gsr::" in n or "multi_tensor" in n:
        print("%-60s calls %5s avg %9.1f us  total %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
