#!/bin/bash
# round 4, GPU call C: the whole GPU test-suite, kernel trace of the bench with the gap attribution, default line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4c
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log | cut -c1-300; grep NATIVE_SHARD $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --no-cpu-baseline --secondary none > $OUT/bench_traced.json 2> $OUT/trace.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) 20 > $OUT/kernel_stats.md 2> $OUT/kernel_stats.err
rm -rf $OUT/trace
tail -28 $OUT/kernel_stats.md | cut -c1-200
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
