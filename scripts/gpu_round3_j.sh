#!/bin/bash
# lanes per workgroup of the blocked Jacobi inner sweep (rebuilds cones.o on the box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3j
mkdir -p $OUT
for T in 512 1024 256; do
  rm -f scs_amd/lib/obj64/cones.o
  make -C scs_amd/csrc ../lib/libscsamd.so EXTRA=-DBJ_INNER_THREADS_OVERRIDE=$T > $OUT/build_$T.log 2>&1 || { tail -5 $OUT/build_$T.log; continue; }
  echo "inner threads $T"
  timeout 600 python scripts/bench_psd_sizes.py --cases 100x32,128x32,256x8,1024x1 2>/dev/null | tee $OUT/psd_T$T.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['k'], d['blocks'], round(d['gpu_ms_per_projection'],2))"
done
