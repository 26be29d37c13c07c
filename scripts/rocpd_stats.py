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

# ---- gaps between consecutive launches (same stream, in start order): where the time
# that is not inside kernels goes.  Only gaps < 100 us count (longer ones are host waits).
ks = c.execute("select start, end, name from kernels order by start").fetchall()
g_all, g_cg = [], []
for (s0, e0, n0), (s1, e1, n1) in zip(ks[:-1], ks[1:]):
    gap = (s1 - e0) / 1000.0
    if 0 <= gap < 100.0:
        g_all.append(gap)
        if ("csr_" in n0 or "k_cg_" in n0) and ("csr_" in n1 or "k_cg_" in n1) and (e0 - s0) > thr * 1000 and (e1 - s1) > thr * 1000:
            g_cg.append(gap)
if g_all:
    import statistics
    span = (ks[-1][1] - ks[0][0]) / 1e6
    busy = sum((e - s) for s, e, _ in ks) / 1e6
    print(f"\n# launch gaps: all {len(g_all)} gaps mean {statistics.mean(g_all):.2f} us median {statistics.median(g_all):.2f} us; "
          f"between active CG-loop kernels {len(g_cg)} gaps mean {statistics.mean(g_cg) if g_cg else 0:.2f} us "
          f"median {statistics.median(g_cg) if g_cg else 0:.2f} us; trace span {span:.1f} ms, inside kernels {busy:.1f} ms")

# ---- where the time OUTSIDE kernels goes ("find the 11 % of trace span outside kernels"): every gap between
# consecutive launches, of any length, attributed to the pair (kernel before -> kernel after) and to a size class
if len(ks) > 1:
    def short(nm):
        nm = nm.replace("scsamd::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        return nm.split("(")[0][:40]
    by_pair, by_class = {}, {"< 5 us": [0, 0.0], "5 - 100 us": [0, 0.0], "100 us - 10 ms": [0, 0.0], ">= 10 ms": [0, 0.0]}
    total_gap = 0.0
    for (s0, e0, n0), (s1, e1, n1) in zip(ks[:-1], ks[1:]):
        gap = (s1 - e0) / 1000.0
        if gap <= 0:
            continue
        total_gap += gap
        cls = "< 5 us" if gap < 5 else ("5 - 100 us" if gap < 100 else ("100 us - 10 ms" if gap < 1e4 else ">= 10 ms"))
        by_class[cls][0] += 1
        by_class[cls][1] += gap
        a = by_pair.setdefault((short(n0), short(n1)), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += gap
        a[2] = max(a[2], gap)
    print(f"\n# time outside kernels: {total_gap/1e3:.1f} ms in {sum(v[0] for v in by_class.values())} gaps")
    print("| gap length | gaps | total ms |")
    print("|---|---|---|")
    for k, v in by_class.items():
        print(f"| {k} | {v[0]} | {v[1]/1e3:.1f} |")
    print("\n| kernel before -> kernel after (largest totals) | gaps | total ms | longest ms |")
    print("|---|---|---|---|")
    for (a, b), v in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| {a} -> {b} | {v[0]} | {v[1]/1e3:.1f} | {v[2]/1e3:.2f} |")
