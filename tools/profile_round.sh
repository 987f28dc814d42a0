#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag>   -- rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats of the bench
# command, HBM FETCH_SIZE / WRITE_SIZE passes and an SQ issue pass (each counter set in its own run, counters only).
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
CMD="bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-train-leg --no-cam-leg --repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o $tag --output-format csv -- python3 $CMD > $out/kt.log 2>&1
CMD1="tools/loop_only.py 60"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o $tag --output-format csv -- python3 $CMD1 > $out/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $out/write -o $tag --output-format csv -- python3 $CMD1 > $out/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $out/sq -o $tag --output-format csv -- python3 $CMD1 > $out/sq.log 2>&1
f() { find $out/$1 -name "$2" | head -1; }
mkdir -p $out/profiles
python3 tools/prof_summary.py --tag $tag --kt "$(f kt ${tag}_kernel_stats.csv)" --fetch "$(f fetch ${tag}_counter_collection.csv)" \
  --write "$(f write ${tag}_counter_collection.csv)" --sq "$(f sq ${tag}_counter_collection.csv)" --cmd "python3 $CMD (kernel trace); $CMD1 (counter passes)" > $out/summary.log 2>&1
# the plain loop (complete lists every iteration: k_preprocess_bin, k_render_fwd<.., 3>): kernel stats + HBM traffic of its own
export LOOP_PLAIN=1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/ktp -o ${tag}plain --output-format csv -- python3 $CMD1 > $out/ktp.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/fetchp -o ${tag}plain --output-format csv -- python3 $CMD1 > $out/fetchp.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $out/writep -o ${tag}plain --output-format csv -- python3 $CMD1 > $out/writep.log 2>&1
unset LOOP_PLAIN
python3 tools/prof_summary.py --tag ${tag}plain --no-latest --kt "$(f ktp ${tag}plain_kernel_stats.csv)" --fetch "$(f fetchp ${tag}plain_counter_collection.csv)" \
  --write "$(f writep ${tag}plain_counter_collection.csv)" --cmd "LOOP_PLAIN=1 python3 $CMD1" > $out/summary_plain.log 2>&1
cp profiles/${tag}* profiles/traffic.json profiles/issue.json $out/profiles/ 2>/dev/null
tail -5 $out/kt.log | cut -c1-600
