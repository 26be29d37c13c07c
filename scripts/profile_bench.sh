#!/bin/bash
# rocprofv3 passes behind profiles/rN_bench_kernel_stats*.md and profiles/rN_pmc_traffic.json (run on the GPU box):
#   1. kernel trace of the default bench command line (no CPU legs / side workloads: they are not kernels of the metric)
#   2. PMC pass FETCH_SIZE             (separate pass: counter slots; never combined with tracing domains other than kernel-trace)
#   3. PMC pass WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${1:-r3}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --secondary none"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python $R/bench.py $ARGS --steps 10 --warmup 5 --no-time-to-eps > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o w -- python $R/bench.py $ARGS --steps 10 --warmup 5 --no-time-to-eps > /dev/null 2> $OUT/pmc_write.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) 20 > $OUT/kernel_stats.md 2> $OUT/kernel_stats.err
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_fetch/*results.db | head -1) 20 > $OUT/pmc_fetch.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_write/*results.db | head -1) 20 > $OUT/pmc_write.md 2>/dev/null
python3 scripts/pmc_traffic_json.py $(ls $OUT/pmc_fetch/*results.db | head -1) $(ls $OUT/pmc_write/*results.db | head -1) $OUT/pmc_traffic.json $OUT/kernel_stats.md "round ${ROUND#r}" > /dev/null 2> $OUT/pmc_json.err
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write   # the databases are large; the summaries are what is kept
head -12 $OUT/kernel_stats.md | cut -c1-200; cat $OUT/pmc_traffic.json
