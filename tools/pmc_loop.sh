#!/bin/bash
# usage (GPU box): tools/pmc_loop.sh <tag> <counter list...>  -- one rocprofv3 counter pass (counters only) over the pure native
# loop (tools/loop_only.py); prints per-kernel per-launch averages of our kernels and keeps them in gpurun_out/pmc_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 tools/loop_only.py 40 > gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY | tee gpurun_out/pmc_$tag.txt
import csv, sys, collections, glob
csv.field_size_limit(sys.maxsize)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
f = glob.glob("gpurun_out/pmc_$tag/**/p_counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gsr::" not in n: continue
    k = n.split("gsr::")[1].split("(")[0].split("<")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k[:24].ljust(24), {kk: round(v / cnt[k][kk]) for kk, v in d.items()})
PY
