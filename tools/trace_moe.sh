#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_moe; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_mixtral.py --layers 2 --iters 3 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4,6,7 "$f" | cut -c1-150 | head -14
tail -1 $OUT/run.log | cut -c1-400
