#!/bin/bash
# usage (GPU box): tools/profile_r04.sh <tag> <name> <scene|train> [plain]
#   rocprofv3 passes behind profiles/<tag>_<name>_*: kernel trace + stats, FETCH_SIZE, WRITE_SIZE and an SQ issue pass (each counter
#   set in its own run, counters only) of ONE workload: the native loop on a named scene (tools/loop_only.py, SCENE=<scene>,
#   `plain` = complete lists in every iteration) or the train.py-style step at 1.5 M Gaussians (tools/train_only.py).
tag=$1; name=$2; what=$3; plain=$4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_${tag}_$name; mkdir -p $out
if [ "$what" = "train" ]; then CMD="tools/train_only.py 40"; ITERS="--iters 43"; else CMD="tools/loop_only.py 60"; export SCENE=$what; ITERS=""; fi
if [ "$plain" = "plain" ]; then export LOOP_PLAIN=1; fi
t=${tag}_$name
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o $t --output-format csv -- python3 $CMD > $out/kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o $t --output-format csv -- python3 $CMD > $out/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $out/write -o $t --output-format csv -- python3 $CMD > $out/write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $out/sq -o $t --output-format csv -- python3 $CMD > $out/sq.log 2>&1
f() { find $out/$1 -name "$2" | head -1; }
python3 tools/prof_summary.py --tag $t --no-latest $ITERS --kt "$(f kt ${t}_kernel_stats.csv)" --fetch "$(f fetch ${t}_counter_collection.csv)" \
  --write "$(f write ${t}_counter_collection.csv)" --sq "$(f sq ${t}_counter_collection.csv)" \
  --cmd "SCENE=$SCENE LOOP_PLAIN=$LOOP_PLAIN python3 $CMD" > $out/summary.log 2>&1
mkdir -p $out/profiles; cp profiles/${t}_* $out/profiles/ 2>/dev/null
tail -2 $out/kt.log | cut -c1-300
unset LOOP_PLAIN SCENE
