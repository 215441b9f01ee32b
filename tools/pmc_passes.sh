#!/bin/bash
# Stall / occupancy counters of one run_config shape, one rocprofv3 --pmc pass per counter group (SQ: 8 slots, TA / TCP: few).
#   usage (through gpurun, from the repo root): tools/pmc_passes.sh <tag> "<configs>" [run_config options]
set -u
TAG=$1; CONFIGS=$2; shift 2
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY TD_TD_BUSY_sum TD_TC_STALL_sum"
)
for C in $CONFIGS; do
  i=0
  for P in "${PASSES[@]}"; do
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/$C/p$i" -- python3 "$REPO/tools/run_config.py" --config $C --iters 3 --warmup 1 "$@" > "$OUT/$C.p$i.log" 2>&1 || echo "pass $i failed for $C"
    i=$((i+1))
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for cdir in sorted(glob.glob(os.path.join(out, "*/"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(cdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "spmv::" not in k or "fill" in k or "count" in k or "rows_kernel" in k or "partition" in k: continue
            acc[k.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", os.path.basename(cdir.rstrip("/")))
    for k, cs in acc.items():
        if not any("blk_" in k or "nat" in k or "sell" in k or "csr" in k for _ in [0]): continue
        print(" ", k)
        for c, v in sorted(cs.items()):
            v = v[len(v) // 2:]   # later dispatches: the timed launches
            print(f"    {c:42s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
