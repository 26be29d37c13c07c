#!/bin/bash
# blocked Jacobi schedules: every pair once per sweep (within pass + cross steps) vs full subproblem sweeps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3m
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_cones_shim_gpu.py tests/test_f32_gpu.py tests/test_golden_gpu.py "tests/test_scale_parity_gpu.py::test_sdp_with_blocks_beyond_the_lds_path_matches_reference_exact_cg" -q --timeout 600 ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for c in 1 0; do
  echo "SCS_AMD_PSD_CROSS=$c"
  SCS_AMD_PSD_CROSS=$c SCS_AMD_DEBUG=1 timeout 600 python scripts/bench_psd_sizes.py --cases 128x32,100x32,128x32,200x16,256x8,512x2,1024x1 2> $OUT/dbg_$c.err | tee $OUT/psd_cross$c.jsonl | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l: d=json.loads(l); print(d['k'], d['blocks'], round(d['gpu_ms_per_projection'],2), d['psd_unconverged'])"
  grep "psd_big" $OUT/dbg_$c.err | tail -2
done
