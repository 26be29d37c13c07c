#!/bin/bash
# round 4, call T: new PSD tests, the bench-line test, then the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4t
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_cones_shim_gpu.py tests/test_bench_gpu.py -m gpu -q -x --timeout 1200 -p no:cacheprovider --durations=5 ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python -c "
import json
d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','ms_per_step','us_per_cg_iter') if k in d}); print(d['roofline']); print(d['secondary'].get('psd_large_blocks'))"
