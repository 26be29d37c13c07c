#!/bin/bash
# round 6: SpMV sweeps behind profiles/r6_spmv_midsize.md and r6_chain_home.md (scripts/bench_spmv_modes.py, one JSON line per case and mode)
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-sweep}
mkdir -p $OUT
cd $R
free -g | head -2; nproc
python scripts/bench_spmv_modes.py --cases 1000000:f64:0 --modes auto+ro0,auto --iters 30 > $OUT/sweep_headline.jsonl 2> $OUT/sweep_headline.err
python scripts/bench_spmv_modes.py --cases 200000:f64:0 --modes auto+ro0,pipe2+ro0,wpc16+nnz512+ro0,wpc16+nnz512+pipe2+ro0,wpc32+nnz256+ro0,wpc12+nnz700+ro0,auto,pipe2,wpc16+nnz512+pipe2 --iters 40 > $OUT/sweep_2e5.jsonl 2> $OUT/sweep_2e5.err
python scripts/bench_spmv_modes.py --cases 100000:f64:0 --modes auto+ro0,pipe2+ro0,wpc16+nnz512+ro0,wpc16+nnz512+pipe2+ro0,auto,pipe2 --iters 40 > $OUT/sweep_1e5.jsonl 2> $OUT/sweep_1e5.err
python scripts/bench_spmv_modes.py --cases 4000000:f32:0 --modes auto+ro0,auto --iters 20 > $OUT/sweep_f32.jsonl 2> $OUT/sweep_f32.err
for f in headline 2e5 1e5 f32; do echo "== $f"; python - $OUT/sweep_$f.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    print("%-34s spmv %7.1f us  frac %.3f  cg-it %7.1f us  init %.2f s  lines %s %s" % (d["mode"], d["spmv_avg_us"], d["frac_of_8TBs"] or 0, d["us_per_cg_iter"] or 0, d.get("scs_init_s",0), [round(v,3) for v in d["numbering"]["lines_per_entry_used"]], d.get("kernels")))
PY
done
