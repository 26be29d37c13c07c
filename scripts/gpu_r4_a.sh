#!/bin/bash
# round 4, GPU call A: full GPU test-suite, PSD order sweep (warm start now covers orders 73..92; A/B against round 3's gate),
# CG vector kernel sweeps (grid cap, chunks in flight, non-temporal policy), the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4a
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
# ---- PSD sizes
timeout 300 python scripts/bench_psd_sizes.py --cases 50x200,64x128,72x100,80x64,92x64,100x32,128x32,256x8,512x2,1024x1 > $OUT/psd_sizes.jsonl 2> $OUT/psd_sizes.err
SCS_AMD_PSD_WARM_KMAX=72 timeout 120 python scripts/bench_psd_sizes.py --cases 80x64,92x64 > $OUT/psd_sizes_warm72.jsonl 2>> $OUT/psd_sizes.err
cut -c1-120 $OUT/psd_sizes.jsonl; cut -c1-120 $OUT/psd_sizes_warm72.jsonl
# ---- CG vector kernels: one short windowed run per setting (us_per_cg_iter of iterations 10..50)
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), window_it_per_s=d['window_it_per_s'])))" >> $OUT/vec_sweep.jsonl
}
: > $OUT/vec_sweep.jsonl
run base X=1
run grid256 SCS_AMD_VEC_MAX_GRID=256
run grid1024 SCS_AMD_VEC_MAX_GRID=1024
run grid2048 SCS_AMD_VEC_MAX_GRID=2048
run upd_unroll2 SCS_AMD_VEC_NT=5
run upd_unroll2_ntpg SCS_AMD_VEC_NT=7
run upd_nt_all SCS_AMD_VEC_NT=3
run upd_plain SCS_AMD_VEC_NT=0
run dir_ntz SCS_AMD_DIR_MODE=1
run dir_ntp SCS_AMD_DIR_MODE=2
run dir_ntzp SCS_AMD_DIR_MODE=3
run dir_unroll2 SCS_AMD_DIR_MODE=4
run dir_unroll2_ntz SCS_AMD_DIR_MODE=5
run base2 X=1
run g1024_u2_d5 SCS_AMD_VEC_MAX_GRID=1024 SCS_AMD_VEC_NT=5 SCS_AMD_DIR_MODE=5
cat $OUT/vec_sweep.jsonl
# ---- the default line
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err; cut -c1-600 $OUT/bench_default.json
