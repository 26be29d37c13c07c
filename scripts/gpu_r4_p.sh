#!/bin/bash
# round 4, call P: PSD kernels after the flag fix (no FLAT accesses) and the look-ahead offsets -- all cone tests, the size sweep, the SDP config
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4p
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests/test_cones_shim_gpu.py tests/test_cones_gpu.py -m gpu -q -x --timeout 1000 -p no:cacheprovider --durations=5 ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for rep in 1 2; do
timeout 300 python scripts/bench_psd_sizes.py --cases 50x200,64x128,72x100,80x64,92x64,100x32,128x32,256x8,512x2,1024x1 > $OUT/psd_sizes_$rep.jsonl 2> $OUT/psd_sizes_$rep.err
cut -c1-100 $OUT/psd_sizes_$rep.jsonl
done
timeout 600 python scripts/bench_sdp.py > $OUT/sdp.json 2> $OUT/sdp.err; tail -c 1500 $OUT/sdp.json
