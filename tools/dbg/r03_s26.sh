#!/bin/bash
# kernel trace of the far split (one frame in flight): where does an iteration at the map's edge spend its time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
for tag in far near; do
  if [ $tag = far ]; then sp="0.3 10"; else sp="0.1 4"; fi
  rm -rf gpurun_out/kt_$tag
  rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$tag -o k --output-format csv -- python3 tools/localize_split.py --frames 24 --in-flight 1 --spread $sp > $o/s26_$tag.log 2>&1
  python3 - $tag <<'PY' > $o/s26_kt_$tag.log
import csv, glob, sys
f = glob.glob("gpurun_out/kt_%s/**/k_kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "gsr::" in n: print("%-40s calls %5s avg %8.1f us  total %8.1f us" % (n.split("gsr::")[1].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  rm -rf gpurun_out/kt_$tag
done
python3 tools/localize_split.py --frames 64 --spread 0.3 10 2>&1 | tail -1 > $o/s26_far8.log
