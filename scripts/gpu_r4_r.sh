#!/bin/bash
# round 4, call R: quick size sweep of the blocked path (two repetitions)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4r
mkdir -p $OUT
cd $R
for rep in 1 2; do
timeout 300 python scripts/bench_psd_sizes.py --cases 100x32,128x32,256x8,512x2,1024x1 2>/dev/null | python -c "
import sys, json
print(' '.join('%dx%d %.4f' % (d['k'], d['blocks'], d['gpu_ms_per_projection']) for d in map(json.loads, (l for l in sys.stdin if l.startswith('{')))))" | tee -a $OUT/sizes.txt
done
timeout 600 python -m pytest tests/test_cones_shim_gpu.py -m gpu -q -x --timeout 500 -p no:cacheprovider 2>&1 | tail -2
