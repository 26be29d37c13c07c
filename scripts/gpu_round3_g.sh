#!/bin/bash
# round 3, GPU call G: pipelined inner sweep of the blocked Jacobi iteration: parity + timing + kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_cones_shim_gpu.py tests/test_golden_gpu.py tests/test_f32_gpu.py "tests/test_scale_parity_gpu.py::test_sdp_with_blocks_beyond_the_lds_path_matches_reference_exact_cg" "tests/test_scale_parity_gpu.py::test_config3_sdp_at_stated_shape_matches_reference_exact_cg" -q --timeout 600 ) > $OUT/pytest_psd.log 2>&1; tail -5 $OUT/pytest_psd.log
timeout 600 python scripts/bench_psd_sizes.py --cases 50x200,100x32,128x32,200x16,256x8,512x2,1024x1 > $OUT/psd_sizes.jsonl 2> $OUT/psd_sizes.err; cat $OUT/psd_sizes.jsonl | cut -c1-160
cd /tmp
export SCS_AMD_GRAPH=0
rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python $R/scripts/bench_psd_sizes.py --cases 1024x1,256x8 --iters 20 > $OUT/psd_traced.out 2> $OUT/psd_traced.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/tr/*results.db | head -1) 0 > $OUT/psd_kernel_stats.md 2>/dev/null
rm -rf $OUT/tr
head -8 $OUT/psd_kernel_stats.md | cut -c1-200
