#!/bin/bash
# waves per CU of the wave-rows layout on banded matrices
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3h
mkdir -p $OUT
for wpc in 4 6 8 12 16; do
  echo "WPC=$wpc"
  SCS_AMD_WR_WPC=$wpc timeout 600 python scripts/bench_locality.py --bands 1024,4096,0 --modes auto 2>/dev/null | tee -a $OUT/locality_wpc$wpc.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['band'], d['pipe'], round(d['spmv_avg_us'],1), round(d['frac_of_8TBs'],3), round(d['us_per_cg_iter'],1))"
done
