"""Per-kernel PMC averages from a rocprofv3 rocpd database.
usage: python scripts/rocpd_pmc.py <results.db> [min_us]"""
import sqlite3
import sys

db = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
c = sqlite3.connect(db)
cols = [d[1] for d in c.execute("pragma table_info('counters_collection')")]
sys.stderr.write(str(cols) + "\n")
q = "select kernel_name, counter_name, value, (end-start)/1000.0 from counters_collection"
try:
    rows = c.execute(q).fetchall()
except Exception as e:
    sys.stderr.write(f"{e}\n")
    rows = []
agg = {}
for name, ctr, val, us in rows:
    if us < thr:
        continue
    a = agg.setdefault((name[:60], ctr), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += val
    a[2] += us
print(f"# PMC per-kernel averages from {db} (dispatches lasting >= {thr} us)")
print("| kernel | counter | dispatches | mean value | mean us |")
print("|---|---|---|---|---|")
for (name, ctr), (n, v, us) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"| {name} | {ctr} | {n} | {v/n:.6g} | {us/n:.1f} |")
