#!/bin/bash
# development helper: PMC counters for the prefill GEMM kernel (rocprofv3, counters only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gemm; mkdir -p $OUT
CASE="${1:-Q4_K,4096,4096,512}"
SETS_FROM=${2:-0}; SETS_TO=${3:-99}; n=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$((n+1)); [ $n -le $SETS_FROM ] && continue; [ $n -gt $SETS_TO ] && continue
  tag=$(echo $set | cut -d' ' -f1)
  echo "== $set"
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/kbench.py --cases $CASE --iters 2 --copies 4 > $OUT/$tag.log 2>&1 || { echo "set failed: $set"; continue; }
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gemm" in k or "prep" in k or "zero_c" in k:
        agg[k[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
done
