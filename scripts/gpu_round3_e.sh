#!/bin/bash
# round 3, GPU call E: persistent PCG loop for latency-bound sizes: bit identity + the tests that now run through it + timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_linsys_gpu.py tests/test_solve_gpu.py tests/test_golden_gpu.py tests/test_conformance_gpu.py tests/test_fuzz_parity_gpu.py tests/test_python_api_gpu.py tests/test_csv_log_gpu.py tests/test_dlong_gpu.py -q --timeout 900 ) > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
for pz in 0 1; do
  SCS_AMD_PERSIST=$pz timeout 600 python scripts/bench_small.py --sizes 1000,3000,10000,20000,40000 > $OUT/small_persist$pz.jsonl 2>/dev/null
  python - <<PY
import json
for l in open("$OUT/small_persist$pz.jsonl"):
    d=json.loads(l); print("PERSIST=$pz n",d["n"],"nnz",d["nnz"],"it/s",d["admm_it_per_s"],"us/cg",d["us_per_cg_it"],"cg/admm",d["cg_its_per_admm"])
PY
done
