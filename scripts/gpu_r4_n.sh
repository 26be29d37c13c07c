#!/bin/bash
# round 4, call N: the fused step of the blocked Jacobi iteration (k_bj_fused) -- PSD parity tests, then the size sweep with and without it
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4n
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_cones_shim_gpu.py -m gpu -q -x --timeout 800 -p no:cacheprovider --durations=5 -k "psd or PSD or order or block" ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for rep in 1 2; do
SCS_AMD_DEBUG=1 timeout 300 python scripts/bench_psd_sizes.py --cases 100x32,128x32,256x8,512x2,1024x1 > $OUT/psd_sizes_$rep.jsonl 2> $OUT/psd_sizes_$rep.err
SCS_AMD_DEBUG=1 SCS_AMD_PSD_FUSED=0 timeout 300 python scripts/bench_psd_sizes.py --cases 100x32,128x32,256x8,512x2,1024x1 > $OUT/psd_sizes_two_launches_$rep.jsonl 2> $OUT/psd_sizes_two_launches_$rep.err
echo "== fused"; cut -c1-100 $OUT/psd_sizes_$rep.jsonl; echo "== two launches"; cut -c1-100 $OUT/psd_sizes_two_launches_$rep.jsonl
done
grep "psd_big" $OUT/psd_sizes_1.err | awk '{print $NF, $(NF-1), $(NF-2)}' | sort | uniq -c | sort -rn | head -20
echo; grep "psd_big" $OUT/psd_sizes_two_launches_1.err | awk '{print $NF, $(NF-1), $(NF-2)}' | sort | uniq -c | sort -rn | head -20
