set -u
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_rand
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPMV_HIP_SLAB_KIB=${SLAB:-512} SPMV_HIP_BLOCK_ROWS=8192 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/tools/quick_bench.py --kind random --methods 4,6 --iters 3 > $OUT/fetch.log 2>&1
SPMV_HIP_SLAB_KIB=${SLAB:-512} SPMV_HIP_BLOCK_ROWS=8192 timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/hit -- python3 $REPO/tools/quick_bench.py --kind random --methods 4,6 --iters 3 > $OUT/hit.log 2>&1
cd $REPO
python3 - <<'PY'
import csv,glob,collections
for sub in ("fetch","hit"):
    for f in glob.glob(f"gpurun_out/pmc_rand/{sub}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "blk_kernel" in k or "csr5_kernel" in k:
                acc[(k[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(sub, k, len(v), sum(v)/len(v))
PY
