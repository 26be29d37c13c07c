"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table.
usage: python scripts/rocpd_stats.py <results.db> [min_us_for_active]"""
import sqlite3
import sys

db = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
c = sqlite3.connect(db)
rows = c.execute("select name, (end-start)/1000.0, grid_x, workgroup_x, vgpr_count, lds_size from kernels").fetchall()
agg = {}
for name, us, gx, wx, vg, lds in rows:
    a = agg.setdefault(name, dict(n=0, tot=0.0, mn=1e30, mx=0.0, act_n=0, act_tot=0.0, vgpr=vg, lds=lds))
    a["n"] += 1
    a["tot"] += us
    a["mn"] = min(a["mn"], us)
    a["mx"] = max(a["mx"], us)
    if us >= thr:
        a["act_n"] += 1
        a["act_tot"] += us
tot = sum(a["tot"] for a in agg.values())
print(f"# kernel-trace summary of {db}; 'active' = launches lasting >= {thr} us (launches enqueued past PCG")
print("# convergence return immediately and would otherwise dilute the average)")
print("| kernel | calls | total ms | % | avg us | min us | max us | active calls | active avg us | vgpr | lds B |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
    short = name if len(name) < 70 else name[:67] + "..."
    aavg = a["act_tot"] / a["act_n"] if a["act_n"] else 0.0
    print(f"| {short} | {a['n']} | {a['tot']/1e3:.2f} | {100*a['tot']/tot:.1f} | {a['tot']/a['n']:.2f} | {a['mn']:.2f} | "
          f"{a['mx']:.2f} | {a['act_n']} | {aavg:.2f} | {a['vgpr']} | {a['lds']} |")
