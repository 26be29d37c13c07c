#!/bin/bash
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r6c7
mkdir -p $OUT
cd $R
bash scripts/r6_reorder_ab.sh r6c7 1000000:f64:0 "auto auto+rh2" 2>&1 | grep -v "k_rescale\|k_cg_\|^#\|^| kernel\|^|---" | cut -c1-250
for ro in 0 1; do
  if [ $ro = 0 ]; then export SCS_AMD_REORDER=0; else unset SCS_AMD_REORDER; fi
  SCS_BENCH_DETAIL=$OUT/bench_ro$ro.detail.json python bench.py --no-cpu-baseline --secondary none > $OUT/bench_ro$ro.json 2> $OUT/bench_ro$ro.err
  python - $OUT/bench_ro$ro.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","iters_to_eps","time_to_eps_s","us_per_cg_iter","cg_its_per_admm_iter","setup_s")}, d["roofline"].get("frac"), d["roofline"].get("avg_launch_us"), d["roofline"].get("kernel"))
PY
done
