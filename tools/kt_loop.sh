#!/bin/bash
# usage (GPU box): tools/kt_loop.sh [iters]  -- rocprofv3 kernel trace of the pure native loop; prints per-kernel average durations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_loop; rocprofv3 --kernel-trace --stats -d gpurun_out/kt_loop -o k --output-format csv -- python3 tools/loop_only.py ${1:-60} > gpurun_out/kt_loop.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/kt_loop/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "gsr::" in n: print("%-40s calls %5s avg %8.1f us  total %8.1f us" % (n.split("gsr::")[1].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
