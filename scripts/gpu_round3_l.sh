#!/bin/bash
# update kernel of the blocked Jacobi iteration: lanes per workgroup (rebuilds cones.o on the box) + parity
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3l
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_cones_shim_gpu.py tests/test_f32_gpu.py "tests/test_scale_parity_gpu.py::test_sdp_with_blocks_beyond_the_lds_path_matches_reference_exact_cg" -q --timeout 600 ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for T in 512 256 1024; do
  rm -f scs_amd/lib/obj64/cones.o
  make -C scs_amd/csrc ../lib/libscsamd.so EXTRA=-DBJ_UPD_THREADS_OVERRIDE=$T > $OUT/build_$T.log 2>&1 || { tail -5 $OUT/build_$T.log; continue; }
  echo "update threads $T"
  timeout 600 python scripts/bench_psd_sizes.py --cases 128x32,128x32,256x8,1024x1 2>/dev/null | tee $OUT/psd_T$T.jsonl | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l: d=json.loads(l); print(d['k'], d['blocks'], round(d['gpu_ms_per_projection'],2))"
done
