#!/bin/bash
# rocprofv3 passes behind profiles/rN_bench_kernel_stats*.md and profiles/rN_pmc_traffic.json (run on the GPU box):
#   1. kernel trace of the default bench command line (no CPU legs / side workloads: they are not kernels of the metric)
#   2. PMC pass FETCH_SIZE             (separate pass: counter slots; never combined with tracing domains other than kernel-trace)
#   3. PMC pass WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1 # A/B script: measurement variants of scs_amd/csrc/options.h are set through the environment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${1:-r3}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --secondary none"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python $R/bench.py $ARGS --steps 10 --warmup 5 --no-time-to-eps > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o w -- python $R/bench.py $ARGS --steps 10 --warmup 5 --no-time-to-eps > /dev/null 2> $OUT/pmc_write.err
#   4. PMC pass L1 -> L2 read requests (how many of the 1e7 gathers per product miss the CU's L1), for the library's choice and for the plain kernel
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_l1 -o l -- python $R/bench.py $ARGS --steps 10 --warmup 5 --no-time-to-eps > /dev/null 2> $OUT/pmc_l1.err
SCS_AMD_WR_LOCKSTEP=0 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_l1_plain -o l -- python $R/bench.py $ARGS --steps 10 --warmup 5 --no-time-to-eps > /dev/null 2> $OUT/pmc_l1_plain.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) 20 > $OUT/kernel_stats.md 2> $OUT/kernel_stats.err
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_fetch/*results.db | head -1) 20 > $OUT/pmc_fetch.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_write/*results.db | head -1) 20 > $OUT/pmc_write.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_l1/*results.db | head -1) 20 > $OUT/pmc_l1.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_l1_plain/*results.db | head -1) 20 > $OUT/pmc_l1_plain.md 2>/dev/null
python3 scripts/pmc_traffic_json.py $(ls $OUT/pmc_fetch/*results.db | head -1) $(ls $OUT/pmc_write/*results.db | head -1) $OUT/pmc_traffic.json $OUT/kernel_stats.md "round ${ROUND#r}" > /dev/null 2> $OUT/pmc_json.err
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_l1 $OUT/pmc_l1_plain   # the databases are large; the summaries are what is kept
head -12 $OUT/kernel_stats.md | cut -c1-200; cat $OUT/pmc_traffic.json; head -8 $OUT/pmc_l1.md; head -8 $OUT/pmc_l1_plain.md
