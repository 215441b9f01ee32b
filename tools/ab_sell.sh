#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
for C in 4 2; do
for T in 0 32; do
  timeout -k 10 200 python3 tools/run_config.py --config $C --method 5 --iters 10 --opt sell_long_thr=$T 2>&1 | grep "RUNCONFIG\|rror" | python3 -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('RUNCONFIG'): print(l.strip()[:300]); continue
    d=json.loads(l[10:]); print(d['config'], d['options'], d['kernel'], 'ms_min', d['ms_min'], 'stored/nnz', round(d['stored_nnz']/d['nnz'],3), 'frac_alg', d['frac_alg'], 'frac_moved', d['frac_moved'], 'inspect', d['inspect_ms'])
"
done
done
