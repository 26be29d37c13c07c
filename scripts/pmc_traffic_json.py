"""Turn the two rocprofv3 PMC passes of scripts/profile_bench.sh into profiles/<round>_pmc_traffic.json (the file bench.py's static
`roofline.traffic` field is read from).   usage: python scripts/pmc_traffic_json.py <fetch.db> <write.db> <out.json> <kernel_stats.md> <label>
FETCH_SIZE on gfx950 tallies 128-byte requests at 64 B for wide streaming reads (MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE KB;
WRITE_SIZE is used as is; TCC_MISS x 128 B is the cross-check."""
import json
import re
import sqlite3
import sys

fetch_db, write_db, out, stats_md, label = sys.argv[1:6]


def per_kernel(db, min_us=20.0):
    c = sqlite3.connect(db)
    agg = {}
    for name, ctr, val, us in c.execute("select kernel_name, counter_name, value, (end-start)/1000.0 from counters_collection"):
        m = re.search(r"csr_wave(?:_lockstep)?_kernel<(\d+)", name)  # the plain / pipelined kernel or the lockstep instantiation
        if us < min_us or not m:
            continue
        a = agg.setdefault((int(m.group(1)), ctr), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += val
        a[2] += us
    return {k: (v[1] / v[0], v[2] / v[0], v[0]) for k, v in agg.items()}


f, w = per_kernel(fetch_db), per_kernel(write_db)
kern = {}
for epi, nm, alg in ((1, "csr_wave[_lockstep]_kernel<DIV> (A, m rows)", 152000004), (2, "csr_wave[_lockstep]_kernel<GP> (A', n rows)", 148000004)):
    fs, us, cnt = f[(epi, "FETCH_SIZE")]
    ws = w[(epi, "WRITE_SIZE")][0]
    hit, miss = w[(epi, "TCC_HIT_sum")][0], w[(epi, "TCC_MISS_sum")][0]
    kern[nm] = dict(FETCH_SIZE_KB=fs, WRITE_SIZE_KB=ws, TCC_HIT=hit, TCC_MISS=miss, hbm_bytes_per_launch=int(2 * fs * 1024 + ws * 1024),
                    tcc_miss_x_128B=int(miss * 128), avg_us_profiled=us, launches=cnt, l2_hit_rate=hit / (hit + miss))
vals = list(kern.values())
prof_us = None
try:
    rows = [l for l in open(stats_md) if re.search(r"csr_wave(?:_lockstep)?_kernel<[12],", l) and len(l.split("|")) >= 12]  # (not the gap table's rows)
    prof_us = sum(float(r.split("|")[9]) for r in rows) / len(rows)  # "active avg us" column of scripts/rocpd_stats.py
except Exception:
    pass
json.dump({
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE ; rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, "
              f"scripts/profile_bench.sh) on `python bench.py --no-cpu-baseline --secondary none --steps 10 --warmup 5 --no-time-to-eps`, headline config, {label}",
    "correction": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 B -> read bytes = 2 x FETCH_SIZE KB; WRITE_SIZE as is; cross-check TCC_MISS_sum x 128 B",
    "kernels": kern,
    "hbm_bytes_per_launch_mean": int(sum(v["hbm_bytes_per_launch"] for v in vals) / len(vals)),
    "algorithmic_bytes_per_launch_mean": 150000004,
    "kernel_avg_us_profiled": prof_us,
    "kernel_stats_source": "rocprofv3 --kernel-trace --stats of the bench command in the same script run (active launches; copied to profiles/ as <round>_bench_kernel_stats.md)",
}, open(out, "w"), indent=1)
print(open(out).read())
