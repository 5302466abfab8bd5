#!/bin/bash
# Round evidence: PMC counters of the north-star kernels (rocprofv3, counters only, one set per run), summarised as JSON:
#   prefill GEMM 4096x4096x512 (int8 body) and 14336x4096x512, the Q8_0 batch body, decode GEMV 4096x4096 and 4096x14336 (Q4_K, Q6_K)
# usage (through gpurun): bash tools/pmc_round.sh  ->  gpurun_out/pmc_round.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_round; rm -rf $OUT; mkdir -p $OUT
CASES="Q4_K,4096,4096,512;Q4_K,14336,4096,512;Q4_K,4096,4096,1;Q4_K,4096,14336,1;Q6_K,4096,14336,1;Q8_0,4096,4096,512"
n=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s$n -- python3 tools/kbench.py --cases "$CASES" --iters 2 --copies 8 > $OUT/s$n.log 2>&1 || echo "set failed: $set"
done
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_lw_kernel" in k:
            key = "gemm_lw " + ("128x64" if "Li2E" in k.split("gemm_mats")[0] else "128x128")
        elif "gemm_ks_kernel" in k:
            key = "gemm_ks 128x64 (4096x4096x512)"
        elif "gemm_kr_kernel" in k:
            key = "gemm_kr 256x128 (14336x4096x512)"
        elif "gemm_i8_kernel" in k:
            key = "gemm_i8 128x64 (4096x4096x512)"
        elif "gemm_lf_q80_kernel" in k:
            key = "gemm_lf_q80 128x64 (Q8_0 4096x4096x512)"
        elif "gemv_kq_kernel" in k and "q4k" in k:
            key = "gemv_q4k " + ("k<=4096" if ", 16, 1," in k or "Li16ELi1E" in k else "deep-k")
        elif "gemv_kq_kernel" in k and "q6k" in k:
            key = "gemv_q6k deep-k (4096x14336)"
        elif "prep_scaled" in k:
            key = "prep_scaled"
        elif "prep_i8" in k:
            key = "prep_i8"
        elif "prep_lf" in k:
            key = "prep_lf"
        else:
            continue
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    c = {n: sum(v) / len(v) for n, v in d.items()}
    c["launches_sampled"] = len(next(iter(d.values())))
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0  # the counter sums the 8 XCDs (MI355X_MICROARCH.md, DVFS)
    if cyc and "SQ_INSTS_MFMA" in c:  # (GRBM_GUI_ACTIVE reads high on dispatches shorter than ~0.3 ms: a LOWER bound of the utilisation)
        c["kernel_cycles"] = cyc
        c["mfma_pipe_utilisation"] = c["SQ_INSTS_MFMA"] * 32.0 / 1024.0 / cyc  # 32 cycles per 32x32x16 MFMA, 1024 SIMDs
    if "FETCH_SIZE" in c:
        c["hbm_bytes"] = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024  # gfx950: FETCH_SIZE counts half of a wide read
    res[k] = {n: (round(v, 4) if v < 100 else round(v, 1)) for n, v in c.items()}
json.dump(res, open("gpurun_out/pmc_round.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/s*/*/*.csv
