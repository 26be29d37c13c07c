#!/bin/bash
# phase clocks of k_psd_jacobi on configs[2] (200 blocks of 50 x 50, warm started): the instrumented variant
# (scripts/build_variant.sh psdclk -DSCSAMD_PSD_CLOCKS, built beforehand -- it travels with the snapshot) swapped in for the run,
# pipelined step (SCS_AMD_PSD_PIPE=1, shipped) and two-phase step (=0) back to back.  Output: gpurun_out/<tag>/psd_clocks.md
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1 # A/B script: measurement variants of scs_amd/csrc/options.h are set through the environment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-psdclk}
mkdir -p $OUT
cd $R
cp scs_amd/lib/libscsamd.so /tmp/lib_shipped.so
cp scs_amd/lib_var/${3:-psdclk}/libscsamd.so scs_amd/lib/libscsamd.so
: > $OUT/psd_clocks.md
for pipe in ${4:-1 0}; do
  SCS_AMD_PSD_PIPE=$pipe timeout 300 python scripts/bench_sdp.py ${2:-} > $OUT/psdclk_pipe$pipe.log 2>&1
  python - $OUT/psdclk_pipe$pipe.log $pipe >> $OUT/psd_clocks.md <<'PY'
import sys, re
rows = []
txt = open(sys.argv[1]).read()  # (device printf and the host's prints share the pipe: lines can run into each other -> patterns, not line splits)
for m in re.finditer(r"PSDCLK pipe (\d+) k (\d+) unpack_warm (\d+) fro (\d+) sweeps (\d+) tail (\d+) nsweep (\d+) steps (\d+) rot_steps (\d+)", txt):
    rows.append(dict(zip(("pipe", "k", "unpack_warm", "fro", "sweeps", "tail", "nsweep", "steps", "rot_steps"), map(float, m.groups()))))
cone = [l.strip() for l in txt.splitlines() if l.startswith("cone:")]
wv = {}
for m in re.finditer(r"PSDWAVE (\w+) work (\d+) wait (\d+)", txt):
    d = wv.setdefault(m.group(1), [0.0, 0.0, 0])
    d[0] += float(m.group(2)); d[1] += float(m.group(3)); d[2] += 1
if not rows:
    print("pipe", sys.argv[2], ": no PSDCLK lines"); sys.exit(0)
n = len(rows)
avg = {k: sum(r[k] for r in rows) / n for k in rows[0]}
tot = avg["unpack_warm"] + avg["fro"] + avg["sweeps"] + avg["tail"]
form = {'1': 'pipelined step, look-ahead', '2': 'pipelined step, signal form', '0': 'two-phase step'}[sys.argv[2]]
print(f"| SCS_AMD_PSD_PIPE={sys.argv[2]} ({form}) | launches {n} | clocks per launch {tot:.0f} "
      f"| unpack+warm {avg['unpack_warm']:.0f} | fro {avg['fro']:.0f} | sweeps {avg['sweeps']:.0f} ({100*avg['sweeps']/tot:.0f} %) | W, WW', pack {avg['tail']:.0f} "
      f"| sweeps per launch {avg['nsweep']:.2f} | steps {avg['steps']:.1f} | rotating steps {avg['rot_steps']:.1f} "
      f"| clocks per rotating step {avg['sweeps']/max(avg['rot_steps'],1):.0f} | {cone[0] if cone else ''} |")
for name, d in wv.items():
    if d[2]:
        print(f"|   {name} wave, per step: work before the barrier {d[0]/d[2]/max(avg['steps'],1):.0f} clocks, waiting at the barrier {d[1]/d[2]/max(avg['steps'],1):.0f} |")
PY
done
cp /tmp/lib_shipped.so scs_amd/lib/libscsamd.so
cat $OUT/psd_clocks.md
