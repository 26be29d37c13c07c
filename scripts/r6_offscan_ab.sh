#!/bin/bash
# round 6: PSD orders > 92 -- the pass over the matrix that decides whether another sweep would rotate anything (psd_offscan = 1,
# shipped) against the closing sweep that rotates nothing (0 = rounds 2-5); parity tests first, then ms per projection, variants round robin
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-offscan_ab}
CASES=${2:-100x32,128x32,200x16,256x8,512x4,1024x1}
ITERS=${3:-40}
mkdir -p $OUT
cd $R
( time timeout 1200 python -m pytest tests/test_cones_shim_gpu.py tests/test_scale_parity_gpu.py tests/test_golden_gpu.py -m gpu -q -p no:cacheprovider -k "psd or sdp or blocked or big or golden" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for rep in ${REPS:-1 2}; do for v in ${VARIANTS:-1_1_1 0_1_1}; do # offscan_grid_prologue
  set -- ${v//_/ }
  echo -n "psd_offscan=$1 psd_grid=$2 psd_prologue=${3:-1}  "; SCS_AMD_PSD_OFFSCAN=$1 SCS_AMD_PSD_GRID=$2 SCS_AMD_PSD_PROLOGUE=${3:-1} python scripts/bench_psd_sizes.py --cases $CASES --iters $ITERS 2>/dev/null | python -c "
import sys, json
print(' '.join('%dx%d %.4f (%.0e)' % (d['k'], d['blocks'], d['gpu_ms_per_projection'], d['max_err_vs_numpy_eigh']) for d in map(json.loads, (l for l in sys.stdin if l.startswith('{')))))"
done; done
