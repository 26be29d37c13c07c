#!/bin/bash
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-sweep3}
mkdir -p $OUT
cd $R
python scripts/bench_spmv_modes.py --cases ${3:-1000000:f64:0} --modes ${2:-auto+rh4,auto+rh4+rs32,auto+rh4+rs128,auto+rh4+rs1024,auto+rh4+rb32,auto+rh4+rb8,auto+rh4+rb64+rs64,auto} --iters 30 > $OUT/sweep.jsonl 2> $OUT/sweep.err
python - $OUT/sweep.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    print("%-34s spmv %7.1f us  frac %.3f  cg-it %7.1f us  cg/admm %.1f init %.2f s  lines %s" % (d["mode"], d["spmv_avg_us"], d["frac_of_8TBs"] or 0, d["us_per_cg_iter"] or 0, d["cg_its_per_admm_iter"], d.get("scs_init_s",0), [round(v,3) for v in d["numbering"]["lines_per_entry_used"]]))
PY
