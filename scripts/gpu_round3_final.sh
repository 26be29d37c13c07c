#!/bin/bash
# round 3, verification call: whole GPU suite, smoke, the default bench line, the rocprofv3 passes of the bench command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3final
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
( time timeout 1500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("value", d["value"], "us/cg", d["us_per_cg_iter"], "frac", d["roofline"]["frac"], "batch", d["batch"]["admm_iters_per_s"])
print("cpu1", d["cpu_baseline"]["value"], "omp", [(l["cores"], l["value"]) for l in d["cpu_baseline_omp"]["legs"]])
s=d["secondary"]; print("sdp", s["configs2_sdp"]["ms_per_projection"], "fp32", s["configs4_fp32"].get("status"), s["configs4_fp32"].get("time_to_eps_s"), "loc", {k: round(v["roofline"]["frac"], 3) for k, v in s["locality_variant"].items()})
PY
# (scripts/profile_bench.sh r3 ran in an earlier call: profiles/r3_pmc_traffic.json)
