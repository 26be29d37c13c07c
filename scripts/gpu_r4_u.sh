#!/bin/bash
# round 4, call U: final validation -- the whole GPU suite, smoke, the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4u
mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 900 ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','us_per_cg_iter') if k in d}, d['roofline']['frac'], d['secondary']['configs2_sdp']['ms_per_projection'], [c['ms_per_projection'] for c in d['secondary']['psd_large_blocks']['cases']])"
