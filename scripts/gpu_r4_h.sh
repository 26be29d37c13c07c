#!/bin/bash
# round 4, GPU call H: whole GPU suite with the lockstep SpMV as the default, profiles (kernel trace, HBM traffic, L1 misses), default line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4h
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log | cut -c1-300; grep NATIVE_SHARD $OUT/pytest.log
bash scripts/profile_bench.sh r4 > $OUT/profile_bench.log 2>&1
tail -40 $OUT/profile_bench.log | cut -c1-220
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
