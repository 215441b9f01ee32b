#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
for C in 3w 3w-uniform; do
  while IFS= read -r OPTS; do
    timeout -k 10 200 python3 tools/run_config.py --config $C --iters 200 --warmup 20 $OPTS 2>&1 | grep "RUNCONFIG\|rror" | python3 -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('RUNCONFIG'): print(l.strip()[:300]); continue
    d=json.loads(l[10:]); print(d['config'], d['method'], d['options'], d['kernel'], 'ms_min', d['ms_min'], 'ms_mean', d['ms_mean'], 'frac_alg', d['frac_alg'])
"
  done <<'LIST'
--method 3
--method 3 --opt csr5_sigma=4
--method 3 --opt csr5_sigma=16
--method 6
--method 6 --opt csr5_sigma=4
--method 6 --opt csr5_sigma=16
--method 1
--method 1 --opt lanes_per_row=1
--method 1 --opt lanes_per_row=2
--method 5
--method 0
LIST
done
