#!/bin/bash
# round 4, GPU call D: lockstep SpMV experiment, the new full-size exact-CG trajectory test, fault-injection test
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4d
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py $B 2>$OUT/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), frac=d['roofline'].get('frac'), window_it_per_s=d['window_it_per_s'], pobj=d['final']['pobj'])))" >> $OUT/lockstep_sweep.jsonl
}
: > $OUT/lockstep_sweep.jsonl
run base X=1
run lockstep SCS_AMD_WR_LOCKSTEP=1
run order_only SCS_AMD_WR_LOCKSTEP=2
run base2 X=1
run lockstep2 SCS_AMD_WR_LOCKSTEP=1
cat $OUT/lockstep_sweep.jsonl; tail -3 $OUT/err_lockstep.txt
( time timeout 1700 python -m pytest tests/test_fault_injection_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 1600 -p no:cacheprovider --durations=8 ) > $OUT/pytest.log 2>&1
tail -16 $OUT/pytest.log | cut -c1-300
