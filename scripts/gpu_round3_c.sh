#!/bin/bash
# round 3, GPU call C: blocked Jacobi for PSD blocks beyond the LDS path: parity tests + timing against the single-column steps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_cones_shim_gpu.py tests/test_golden_gpu.py tests/test_f32_gpu.py "tests/test_scale_parity_gpu.py::test_sdp_with_blocks_beyond_the_lds_path_matches_reference_exact_cg" -q --timeout 600 ) > $OUT/pytest_psd.log 2>&1; tail -12 $OUT/pytest_psd.log
CASES="100x32,128x32,200x16,256x8,512x2,1024x1"
timeout 600 python scripts/bench_psd_sizes.py --cases $CASES > $OUT/psd_sizes_blocked.jsonl 2> $OUT/psd_sizes_blocked.err; cat $OUT/psd_sizes_blocked.jsonl
SCS_AMD_PSD_BLOCKED=0 timeout 600 python scripts/bench_psd_sizes.py --cases $CASES > $OUT/psd_sizes_columns.jsonl 2> $OUT/psd_sizes_columns.err; cat $OUT/psd_sizes_columns.jsonl
( timeout 600 python -m pytest tests/test_linsys_gpu.py tests/test_cones_exp_pow_gpu.py -q --timeout 600 ) > $OUT/pytest_linsys.log 2>&1; tail -5 $OUT/pytest_linsys.log
timeout 900 python scripts/bench_locality.py > $OUT/locality.jsonl 2> $OUT/locality.err; cat $OUT/locality.jsonl
