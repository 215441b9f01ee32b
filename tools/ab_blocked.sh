#!/bin/bash
# A/B of the row-block x column-slab executor forms (variants, block_rows, slab width) on the no-locality shapes
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CONFIGS=${1:-"2r 3o-uniform"}
for C in $CONFIGS; do
  while IFS= read -r OPTS; do
    timeout -k 10 200 python3 tools/run_config.py --config $C --method 4 --iters 8 $OPTS 2>&1 | grep "RUNCONFIG\|rror" | python3 -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('RUNCONFIG'): print(l.strip()[:200]); continue
    d=json.loads(l[10:]); print(d['config'], d['options'], d['kernel'], 'ms_min', d['ms_min'], 'create_s', d['create_s'], 'inspect_ms', d['inspect_ms'], 'frac_alg', d['frac_alg'], 'frac_moved', d['frac_moved'])
"
  done <<'LIST'
--opt variant=19
--opt variant=20
--opt variant=21
--opt block_rows=4096 --opt variant=19
--opt block_rows=4096 --opt variant=20
--opt block_rows=4096 --opt variant=21
--opt block_rows=2048 --opt variant=20
LIST
done
