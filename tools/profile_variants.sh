#!/bin/bash
# usage (GPU box): tools/profile_variants.sh <tag>   -- the passes of tools/profile_round.sh for the structured variants of S-1M-640
# (speculative loop, one frame; tools/loop_only.py with SCENE=...): profiles/<tag>_<variant>_spec_{kernel_stats,issue_utilisation}.md, _traffic.json
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
f() { find $1 -name "$2" | head -1; }
for v in object:s_1m_640_object walls:s_1m_640_walls room:s_room_640; do
  name=${v%%:*}; export SCENE=${v#*:}
  t=${tag}_${name}_spec
  out=gpurun_out/prof_$t; mkdir -p $out
  CMD1="tools/loop_only.py 60"
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o $t --output-format csv -- python3 $CMD1 > $out/kt.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o $t --output-format csv -- python3 $CMD1 > $out/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $out/write -o $t --output-format csv -- python3 $CMD1 > $out/write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $out/sq -o $t --output-format csv -- python3 $CMD1 > $out/sq.log 2>&1
  python3 tools/prof_summary.py --tag $t --no-latest --kt "$(f $out/kt ${t}_kernel_stats.csv)" --fetch "$(f $out/fetch ${t}_counter_collection.csv)" \
    --write "$(f $out/write ${t}_counter_collection.csv)" --sq "$(f $out/sq ${t}_counter_collection.csv)" --cmd "SCENE=$SCENE python3 $CMD1" > $out/summary.log 2>&1
  mkdir -p $out/profiles; cp profiles/${t}* $out/profiles/ 2>/dev/null
  tail -2 $out/summary.log | cut -c1-300
done
