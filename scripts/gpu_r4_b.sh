#!/bin/bash
# round 4, GPU call B: the tests that failed / are new, x-window prefetch sweep, rocprofv3 passes (bench + blocked PSD), default line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4b
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_fault_injection_gpu.py tests/test_reorder_gpu.py tests/test_linsys_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), frac=d['roofline'].get('frac'), window_it_per_s=d['window_it_per_s'], pobj=d['final']['pobj'])))" >> $OUT/pf_sweep.jsonl
}
: > $OUT/pf_sweep.jsonl
run base X=1
run pf1 SCS_AMD_WR_PF=1
run pf2 SCS_AMD_WR_PF=2
run pf3 SCS_AMD_WR_PF=3
run pf4 SCS_AMD_WR_PF=4
run pf6 SCS_AMD_WR_PF=6
run base2 X=1
cat $OUT/pf_sweep.jsonl
# rocprofv3: bench kernel trace + PMC traffic (scripts/profile_bench.sh writes gpurun_out/prof_r4)
bash scripts/profile_bench.sh r4 > $OUT/profile_bench.log 2>&1
tail -15 $OUT/profile_bench.log | cut -c1-220
bash scripts/profile_psd_blocked.sh > $OUT/profile_psd.log 2>&1
tail -12 $OUT/profile_psd.log | cut -c1-200
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
( time timeout 600 python -m pytest tests/test_bench_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k side_workloads ) > $OUT/pytest_bench.log 2>&1
tail -5 $OUT/pytest_bench.log
