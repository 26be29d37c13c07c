#!/bin/bash
# round 4, GPU call E: lockstep SpMV variants; new PSD tests; the zero-cone-weighted linear solve at the headline size
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4e
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py $B 2>$OUT/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), frac=d['roofline'].get('frac'), window_it_per_s=d['window_it_per_s'])))" >> $OUT/lockstep_sweep.jsonl
}
: > $OUT/lockstep_sweep.jsonl
run base X=1
run ls8_b4 SCS_AMD_WR_LOCKSTEP=1
run ls8_b2 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_BARRIERS=2
run ls8_b1 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_BARRIERS=1
run ls8_b0 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_BARRIERS=0
run ls16_b4 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_WPB=16 SCS_AMD_WR_WPC=16
run ls16_b1 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_LS_WPB=16 SCS_AMD_WR_WPC=16 SCS_AMD_WR_LS_BARRIERS=1
run ls8_b4_wpc16 SCS_AMD_WR_LOCKSTEP=1 SCS_AMD_WR_WPC=16
run base2 X=1
cat $OUT/lockstep_sweep.jsonl
( time timeout 1700 python -m pytest tests/test_cones_shim_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 1600 -p no:cacheprovider --durations=6 -k "psd or weighting or lds_kernel" ) > $OUT/pytest.log 2>&1
tail -14 $OUT/pytest.log | cut -c1-300
