#!/bin/bash
# round 6: what the chain + home numbering does to the two CG products, by rocprofv3 (kernel trace: true per-orientation durations; one PMC
# pass: L1 -> L2 read requests) -- the same workload (scripts/bench_spmv_modes.py, headline case) with the numbering off / on
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-reorder_ab}
CASE=${2:-1000000:f64:0}
MODES=${3:-"auto+ro0 auto auto+pipe0"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in $MODES; do
  tag=$(echo $mode | tr '+' '_')
  rocprofv3 --kernel-trace --stats -d $OUT/tr_$tag -o t -- python $R/scripts/bench_spmv_modes.py --cases $CASE --modes $mode --iters 20 > $OUT/run_$tag.jsonl 2> $OUT/run_$tag.err
  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_$tag -o p -- python $R/scripts/bench_spmv_modes.py --cases $CASE --modes $mode --iters 10 > /dev/null 2> $OUT/pmc_$tag.err
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2_$tag -o p -- python $R/scripts/bench_spmv_modes.py --cases $CASE --modes $mode --iters 10 > /dev/null 2> $OUT/pmc2_$tag.err
  ( cd $R; python3 scripts/rocpd_pmc.py $(ls $OUT/pmc2_$tag/*results.db | head -1) 20 2>/dev/null | head -8 > $OUT/pmc2_$tag.md ); rm -rf $OUT/pmc2_$tag
  ( cd $R; python3 scripts/rocpd_stats.py $(ls $OUT/tr_$tag/*results.db | head -1) 8 > $OUT/stats_$tag.md 2>/dev/null; python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_$tag/*results.db | head -1) 20 2>/dev/null | head -8 > $OUT/pmc_$tag.md )
  rm -rf $OUT/tr_$tag $OUT/pmc_$tag
  echo "== $mode"; head -9 $OUT/stats_$tag.md | cut -c1-170; head -8 $OUT/pmc_$tag.md | cut -c1-170; head -8 $OUT/pmc2_$tag.md | cut -c1-170; cat $OUT/run_$tag.jsonl | cut -c1-400
done
