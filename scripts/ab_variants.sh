#!/bin/bash
# A/B of library variants built by scripts/build_variant.sh on ONE box: swaps scs_amd/lib/libscsamd.so, runs the size sweep, round robin
# usage: scripts/ab_variants.sh "<cases>" <reps> <variant>...
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1 # A/B script: measurement variants of scs_amd/csrc/options.h are set through the environment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ab
mkdir -p $OUT
cd $R
cases=$1; reps=$2; shift 2
cp scs_amd/lib/libscsamd.so /tmp/lib_shipped.so
: > $OUT/ab.txt
for rep in $(seq $reps); do
  for v in "$@"; do
    cp scs_amd/lib_var/$v/libscsamd.so scs_amd/lib/libscsamd.so
    timeout 300 python scripts/bench_psd_sizes.py --cases $cases 2>/dev/null | python -c "
import sys, json
print('%-10s' % '$v', ' '.join('%dx%d %.4f' % (d['k'], d['blocks'], d['gpu_ms_per_projection']) for d in map(json.loads, (l for l in sys.stdin if l.startswith('{')))))" | tee -a $OUT/ab.txt
  done
done
cp /tmp/lib_shipped.so scs_amd/lib/libscsamd.so
