#!/bin/bash
# LDS Jacobi kernel (configs[2]) with grouped operand loads: parity + timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_cones_shim_gpu.py tests/test_golden_gpu.py tests/test_f32_gpu.py tests/test_scale_parity_gpu.py tests/test_fuzz_parity_gpu.py tests/test_conformance_gpu.py -q --timeout 600 ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python scripts/bench_sdp.py 2>&1 | tail -3 | cut -c1-300
timeout 600 python scripts/bench_psd_sizes.py --cases 4x2000,16x500,32x200,50x200,64x128,92x64,100x32,256x8,1024x1 2>/dev/null | tee $OUT/psd_sizes.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['k'], d['blocks'], round(d['gpu_ms_per_projection'],3))"
