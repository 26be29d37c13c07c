#!/bin/bash
# non-temporal cache policy on the streams of k_cg_update (x, r, M; or all of them): whole-solve rate and us per CG iteration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
for nt in 0 1 3 0 1; do
  SCS_AMD_VEC_NT=$nt python bench.py --no-cpu-baseline --secondary none 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('VEC_NT=$nt value', round(d['value'],2), 'us/cg', round(d['us_per_cg_iter'],2), 'spmv us', round(d['roofline']['avg_launch_us'],2), 'pobj', d['final']['pobj'])"
done
