#!/bin/bash
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r6c17
mkdir -p $OUT
cd $R
python scripts/bench_spmv_modes.py --cases 4000000:f32:0 --modes auto+ro0,auto+ro1 --iters 15 > $OUT/sweep_f32.jsonl 2> $OUT/sweep_f32.err
python scripts/bench_spmv_modes.py --cases 200000:f64:0 --modes auto+ro0,auto --iters 40 > $OUT/sweep_2e5.jsonl 2> $OUT/sweep_2e5.err
for f in f32 2e5; do echo "== $f"; python - $OUT/sweep_$f.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    print("%-24s spmv %7.1f us  frac %.3f  cg-it %7.1f us  cg/admm %.1f init %.2f s  lines %s %s" % (d["mode"], d["spmv_avg_us"], d["frac_of_8TBs"] or 0, d["us_per_cg_iter"] or 0, d["cg_its_per_admm_iter"], d.get("scs_init_s",0), [round(v,3) for v in d["numbering"]["lines_per_entry_used"]], d.get("kernels")))
PY
done
bash scripts/gpu_run.sh r6prof2 profile:r6 2>&1 | tail -30
