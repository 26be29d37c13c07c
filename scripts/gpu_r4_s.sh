#!/bin/bash
# round 4, call S: per-launch duration of k_bj_fused (rocprofv3 kernel trace) for library variants
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r4s
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp scs_amd/lib/libscsamd.so /tmp/lib_shipped.so
for v in "$@"; do
  cp scs_amd/lib_var/$v/libscsamd.so scs_amd/lib/libscsamd.so
  rm -rf /tmp/tr_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python scripts/bench_psd_sizes.py --cases 1024x1 > $OUT/$v.log 2>&1
  f=$(find /tmp/tr_$v -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:4]:
    v.sort()
    print(sys.argv[2], k, "calls", len(v), "avg %.2f us" % (sum(v) / len(v)), "median %.2f" % v[len(v) // 2], "p90 %.2f" % v[int(len(v) * 0.9)])
PY
done
cp /tmp/lib_shipped.so scs_amd/lib/libscsamd.so
