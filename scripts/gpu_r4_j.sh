#!/bin/bash
# round 4, GPU call J: stream-ahead inside the lockstep SpMV
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4j
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 800 python scripts/bench_spmv_modes.py --cases 1000000:f64:0,1000000:f64:1024,1000000:f64:65536 --modes auto,ls16a,ls16b1a,ls16,ls16b1 > $OUT/modes.jsonl 2> $OUT/modes.err
grep "^{" $OUT/modes.jsonl | cut -c1-200
