#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   1. kernel trace + stats of the default bench command         -> gpurun_out/prof/stats
#   2. PMC passes (own runs, counters only): FETCH_SIZE, WRITE_SIZE of the decode GEMV + prefill GEMM
# Summaries are written to gpurun_out/prof/*.json|csv; copy what should be judged into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $OUT/bench_profiled.json 2> $OUT/stats.err
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" $OUT/kernel_stats_trimmed.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    w = csv.writer(o)
    w.writerow(rows[0])
    for r in rows[1:]:
        if any(k in r[0] for k in ("gemv", "gemm", "prep", "quantize", "pack", "moe")):  # (gemm_i8_kernel / prep_i8_kernel included)
            r[0] = r[0][:110]
            w.writerow(r)
PY
rm -rf $OUT/stats/*/*kernel_trace.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -- python3 bench.py --steps 1 --warmup 0 --decode 4 --no-graph --no-cpu-baseline --no-extra-configs > $OUT/pmc_$ctr.json 2> $OUT/pmc_$ctr.err
done
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemv_kq_kernel" in k and "q4k_traits" in k:
            agg["gemv_q4k"].append(float(r["Counter_Value"]))
        elif "gemm_kq_kernel" in k and "Li12E" in k:
            agg["gemm_q4k_narrow"].append(float(r["Counter_Value"]))
        elif "gemm_wide_kernel" in k and "Li12E" in k:
            agg["gemm_q4k_wide"].append(float(r["Counter_Value"]))
        elif "gemm_lw_kernel" in k and "Li12E" in k:
            agg["gemm_q4k_lw_" + ("128x64" if "Li2E" in k else "128x128")].append(float(r["Counter_Value"]))
        elif "gemm_ks_kernel" in k:
            agg["gemm_q4k_ks_128x64"].append(float(r["Counter_Value"]))
        elif "gemm_i8_kernel" in k:
            agg["gemm_q4k_i8_128x64"].append(float(r["Counter_Value"]))
        elif "gemm_kr_kernel" in k:
            agg["gemm_q4k_kr_256x128"].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][ctr + "_KB_avg_per_launch"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
for k, d in res.items():
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) reports 1/2 of a wide coalesced read on gfx950 -> x2; WRITE_SIZE exact
    d["hbm_bytes_per_launch"] = int((2 * d.get("FETCH_SIZE_KB_avg_per_launch", 0) + d.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024)
sys.path.insert(0, ".")
import bench  # the profile is quoted by bench.py only for the kernel sources it was measured on
res["kernel_source_sha256"] = bench.kernel_source_hash()
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/pmc_FETCH_SIZE/*/*.csv $OUT/pmc_WRITE_SIZE/*/*.csv
cat $OUT/kernel_stats_trimmed.csv | cut -c1-200
