#!/bin/bash
# usage: tools/pmc_kernels.sh <tag> <counter list...>   -> prints per-kernel averages of our kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 bench.py --steps 5 --warmup 1 --frames-in-flight 1 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, sys, collections
csv.field_size_limit(sys.maxsize)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open("gpurun_out/pmc_$tag/p_counter_collection.csv")):
    n = r["Kernel_Name"]
    if "gsr::" not in n: continue
    k = n.split("gsr::")[1].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k[:40], {kk: round(v / cnt[k][kk]) for kk, v in d.items()})
PY
