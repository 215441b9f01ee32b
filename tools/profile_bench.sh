#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   pass 1  --kernel-trace --stats            per-kernel durations of `python3 bench.py`
#   pass 2  --pmc FETCH_SIZE                  (TCC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass)
#   pass 3  --pmc WRITE_SIZE
#   pass 4  --pmc FETCH_SIZE on build/kbench  calibration: stream_read reads a KNOWN 3.84 GB with the
#                                             same 16 B/lane loads (MI355X_MICROARCH.md "HBM")
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); tools/summarize_profile.py condenses it into
# profiles/<tag>_*.{csv,json} (tracked).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu > "$OUT/fetch.log" 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu > "$OUT/write.log" 2>&1 || echo "write pass failed"
if [ -x "$REPO/build/kbench" ]; then
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib" -- "$REPO/build/kbench" 10000000 1 > "$OUT/calib.log" 2>&1 || echo "calib pass failed"
fi
cd "$REPO"
find "$OUT" -name "*.csv" | head -20
python3 tools/summarize_profile.py "$TAG" || true
