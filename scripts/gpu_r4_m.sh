#!/bin/bash
# round 4, call M: the CU-sorted product (SCS_AMD_WR_CUSORT=1) against the lockstep default -- parity first, then time
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4m
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
SCS_AMD_WR_CUSORT=1 SCS_AMD_DEBUG=1 timeout 600 python -m pytest tests/test_linsys_gpu.py -m gpu -x -q > $OUT/pytest_linsys_cusort.txt 2>&1
tail -5 $OUT/pytest_linsys_cusort.txt
SCS_AMD_WR_CUSORT=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "weighted" > $OUT/pytest_fullsize_cusort.txt 2>&1
tail -3 $OUT/pytest_fullsize_cusort.txt
B="--no-cpu-baseline --secondary none --steps 40 --warmup 10 --no-time-to-eps"
run() { local label=$1; shift
  env "$@" timeout 300 python bench.py $B 2>$OUT/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(json.dumps(dict(label='$label', us_per_cg_iter=d['us_per_cg_iter'], spmv_us=d['roofline'].get('avg_launch_us'), frac=d['roofline'].get('frac'))))" >> $OUT/sweep.jsonl
}
: > $OUT/sweep.jsonl
run auto X=1
run cusort SCS_AMD_WR_CUSORT=1 SCS_AMD_DEBUG=1
run auto2 X=1
run cusort2 SCS_AMD_WR_CUSORT=1
cat $OUT/sweep.jsonl
grep cusort $OUT/err_cusort.txt | head
