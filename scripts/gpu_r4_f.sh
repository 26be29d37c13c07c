#!/bin/bash
# round 4, GPU call F: lockstep SpMV, gather-first pipeline variants
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4f
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py $B 2>$OUT/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), frac=d['roofline'].get('frac'), window_it_per_s=d['window_it_per_s'])))" >> $OUT/lockstep_sweep.jsonl
}
: > $OUT/lockstep_sweep.jsonl
L16="SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_WPB=16 SCS_AMD_WR_WPC=16"
run base X=1
run ls16_b4 $L16 SCS_AMD_WR_LS_BARRIERS=4
run ls16_gf_b4 $L16 SCS_AMD_WR_LS_BARRIERS=12
run ls16_gf_b1 $L16 SCS_AMD_WR_LS_BARRIERS=9
run ls16_gf_b0 $L16 SCS_AMD_WR_LS_BARRIERS=8
run ls8_gf_b4 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_BARRIERS=12
run ls16_gf_b4_again $L16 SCS_AMD_WR_LS_BARRIERS=12
cat $OUT/lockstep_sweep.jsonl; tail -2 $OUT/err_ls16_gf_b4.txt
