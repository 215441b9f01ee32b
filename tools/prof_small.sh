#!/bin/bash
# kernel-level view of a small (launch-latency-scale) case: webbase-style stand-in, Method_Balanced2
set -u
REPO=$PWD; OUT=$REPO/gpurun_out/prof_small; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/tools/quick_bench.py --kind powerlaw --m 1000000 --k 3 --maxlen 4700 --methods 3,6,1 --iters 200 > $OUT/log.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/prof_small/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -3 $OUT/log.txt | cut -c1-200
