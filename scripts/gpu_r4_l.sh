#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4l
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { local label=$1; shift
  env "$@" timeout 200 python bench.py $B 2>$OUT/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), frac=d['roofline'].get('frac'))))" >> $OUT/sweep.jsonl
}
: > $OUT/sweep.jsonl
run auto X=1
run no_subwindow_order SCS_AMD_WR_LS_ORDER=0
run wpb12 SCS_AMD_WR_LS_WPB=12 SCS_AMD_WR_LS_BARRIERS=4
run nt_stream SCS_AMD_WR_LS_BARRIERS=12
run auto2 X=1
cat $OUT/sweep.jsonl
