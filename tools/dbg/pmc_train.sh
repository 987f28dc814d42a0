#!/bin/bash
# HBM write / fetch bytes of the binning kernels at the train config (P = 1.5 M)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/r02/pmc_train_$c -o p --output-format csv -- python3 tools/dbg/train_kernels.py > gpurun_out/r02/pmc_train_$c.log 2>&1
done
python3 - <<PY
import csv, sys, collections, glob
csv.field_size_limit(sys.maxsize)
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02/pmc_train_{c}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gsr::" not in n: continue
            k = n.split("gsr::")[1].split("(")[0][:40]
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)
        print(c, k, "n=%d" % len(v), "min %.0f KiB  median %.0f KiB  max %.0f KiB" % (v[0], v[len(v)//2], v[-1]))
PY
