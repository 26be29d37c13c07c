#!/bin/bash
# round 3, GPU call B: fused update+direction kernel (bit identity, in-situ timing at n = 1e6 and 2e5), the default bench line,
# batch concurrency sweep, rocprofv3 kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_linsys_gpu.py -q -x --timeout 600 -k "fused or small_n" ) > $OUT/pytest_fused.log 2>&1; tail -3 $OUT/pytest_fused.log
for f in 0 1; do
  SCS_AMD_CGFUSE=$f timeout 600 python bench.py --no-cpu-baseline --secondary none > $OUT/bench_fuse$f.json 2> $OUT/bench_fuse$f.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_fuse$f.json"))
print("CGFUSE=$f value", d["value"], "us_per_cg_iter", d["us_per_cg_iter"], "spmv us", d["roofline"].get("avg_launch_us"), "iters", d["iters_to_eps"], "pobj", d["final"]["pobj"])
PY
  SCS_AMD_CGFUSE=$f timeout 300 python scripts/bench_small.py --sizes 20000,200000 > $OUT/small_fuse$f.jsonl 2>/dev/null; cat $OUT/small_fuse$f.jsonl
done
for c in 2 4 8; do
  timeout 600 python bench.py --no-cpu-baseline --secondary batch --n 20000 --steps 5 --warmup 2 --batch-concurrency $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batch']; print('batch concurrency $c', b['admm_iters_per_s'], b['problems_per_s'], b['wall_s'])"
done
( time timeout 1500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 6000 $OUT/bench_default.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --no-cpu-baseline --secondary none > $OUT/bench_traced.json 2> $OUT/trace.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) 20 > $OUT/kernel_stats.md 2> $OUT/kernel_stats.err
rm -rf $OUT/trace
head -24 $OUT/kernel_stats.md
