#!/bin/bash
# development helper: per-kernel durations of the prefill GEMM path (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_gemm; rm -rf $OUT; mkdir -p $OUT
CASES="${1:-Q4_K,4096,4096,512}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/kbench.py --cases "$CASES" --iters 5 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4,6,7 "$f" | cut -c1-160
grep us/launch $OUT/run.log
