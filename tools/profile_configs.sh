#!/bin/bash
# rocprofv3 evidence for every BASELINE config's named kernel (run through gpurun from the repo root):
#   per config: --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc TCC_HIT_sum TCC_MISS_sum
# (separate passes: FETCH_SIZE and WRITE_SIZE do not fit the TCC slots together; PMC passes carry --kernel-trace only).
# Plus the same for `python3 bench.py` (config 2 through the bench itself) and the known-bytes calibration kernels.
# Raw output: gpurun_out/prof_<tag>/ (scratch); tools/summarize_configs.py condenses it into profiles/ (tracked).
#   usage: tools/profile_configs.sh r02 "2 2r 3w 3o 4"      (NOBENCH=1: configs only; KEEP=1: add to an existing output directory)
set -u
TAG=${1:-r02}
CONFIGS=${2:-"2 2h s27 2r 3w 3o 3o-uniform 4"}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
[ -n "${KEEP:-}" ] || rm -rf "$OUT"   # KEEP=1: second call of a split run
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in $CONFIGS; do
  echo "== config $C"; date +%T
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$C/stats" -- python3 "$REPO/tools/run_config.py" --config $C --iters 20 > "$OUT/$C.stats.log" 2>&1 || echo "stats pass failed for $C"
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/$C/fetch" -- python3 "$REPO/tools/run_config.py" --config $C --iters 4 --warmup 1 > "$OUT/$C.fetch.log" 2>&1 || echo "fetch pass failed for $C"
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/$C/write" -- python3 "$REPO/tools/run_config.py" --config $C --iters 4 --warmup 1 > "$OUT/$C.write.log" 2>&1 || echo "write pass failed for $C"
  timeout -k 10 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/$C/hit" -- python3 "$REPO/tools/run_config.py" --config $C --iters 4 --warmup 1 > "$OUT/$C.hit.log" 2>&1 || echo "hit pass failed for $C"
  grep -h RUNCONFIG "$OUT/$C.stats.log" | cut -c1-400
done
if [ -z "${NOBENCH:-}" ]; then
echo "== bench.py"; date +%T
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench/stats" -- python3 "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu > "$OUT/bench.stats.log" 2>&1 || echo "bench stats pass failed"
if [ -x "$REPO/spmv_amd/bin/gbench" ]; then   # built by spmv_amd/build.py from tools/gbench.hip
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/calib/fetch" -- "$REPO/spmv_amd/bin/gbench" calib > "$OUT/calib.log" 2>&1 || echo "calib pass failed"
fi
fi
cd "$REPO"
# then, in the checkout that received gpurun_out/: python3 tools/summarize_configs.py "$TAG"   (rewrites profiles/<tag>_*; not run here,
# so that a later bench.py on this box still finds the committed profiles/traffic_<tag>.json)
