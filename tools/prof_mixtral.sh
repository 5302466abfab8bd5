#!/bin/bash
# development: kernel stats of the Mixtral expert path (2 layers)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_mix; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 tools/bench_mixtral.py --layers 2 --iters 2 > $OUT/out.json 2> $OUT/err.log
f=$(find $OUT/s -name "*kernel_stats.csv" | head -1)
cut -c1-150 "$f" | head -25
rm -rf $OUT/s/*/*kernel_trace.csv
