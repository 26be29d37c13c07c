#!/bin/bash
# round 4, GPU call G: where does the lockstep instantiation pay (sizes, fp32, banded patterns)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4g
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 1200 python scripts/bench_spmv_modes.py --cases 1000000:f64:0,100000:f64:0,200000:f64:0,500000:f64:0,2000000:f64:0,4000000:f32:0,1000000:f64:1024,1000000:f64:65536 --modes plain,ls16,ls16b1,ls8 > $OUT/modes.jsonl 2> $OUT/modes.err
cat $OUT/modes.jsonl | cut -c1-220; tail -3 $OUT/modes.err
