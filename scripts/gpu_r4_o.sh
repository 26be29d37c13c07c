#!/bin/bash
# round 4, call O: phase clocks of the inner sweep (build with EXTRA=-DBJ_PROFILE)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4o
mkdir -p $OUT
cd $R
timeout 300 python scripts/bench_psd_sizes.py --cases 1024x1 > $OUT/a.jsonl 2> $OUT/a.err
grep bj_prof $OUT/a.err | tail -1; cut -c1-120 $OUT/a.jsonl
