#!/bin/bash
# round 6: the signal form of the pipelined PSD step (psd_pipe = 2) against the look-ahead form (1) and the two-phase step (0)
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-psd_ab}
mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_cones_shim_gpu.py tests/test_golden_gpu.py -m gpu -q -p no:cacheprovider -k "psd or pipelined or golden" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for rep in 1 2 3; do for pipe in 1 2 0; do
  echo -n "psd_pipe=$pipe  "; SCS_AMD_PSD_PIPE=$pipe python scripts/bench_psd_sizes.py --cases 16x500,32x200,50x200,64x128,72x100 --iters 40 2>/dev/null | python -c "
import sys, json
print(' '.join('%dx%d %.4f (%.0e)' % (d['k'], d['blocks'], d['gpu_ms_per_projection'], d['max_err_vs_numpy_eigh']) for d in map(json.loads, (l for l in sys.stdin if l.startswith('{')))))"
done; done
for pipe in 1 2; do echo -n "configs2 psd_pipe=$pipe: "; SCS_AMD_PSD_PIPE=$pipe python scripts/bench_sdp.py 2>/dev/null | grep "^cone:"; done
bash scripts/psd_clocks.sh ${1:-psd_ab}_clk "" psdclk "2 1" > /dev/null 2>&1; cut -c1-700 gpurun_out/${1:-psd_ab}_clk/psd_clocks.md
