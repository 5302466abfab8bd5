#!/bin/bash
# development: kernel trace of a short bench run; prints start / duration / gap of the kernels of one decode pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --steps 1 --warmup 1 --decode 8 --no-cpu-baseline --no-extra-configs > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 160 kernels that are decode GEMVs = the tail of the last decode pass
dec = [r for r in rows if "gemv" in r["Kernel_Name"]]
tail = dec[-150:]
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = n[n.find("gemv"):][:64]
    gap = (s - prev_end) if prev_end else 0
    print(f"{short:64s} dur {e - s:6d} ns  gap {gap:6d} ns")
    prev_end = e
PY
rm -rf $OUT/t
